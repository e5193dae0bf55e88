"""Thin ctypes binding of libdabgpu.so (the C ABI in include/dabgpu.h).

Plumbing only: argument marshalling and error translation.  There is no Python or
CPU implementation behind these calls -- if the HIP library is missing or no gfx950
device is visible they raise.  Used by tests/, bench.py and __graft_entry__.py.
"""
import atexit
import ctypes as C
import os
import sys
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DABGPU_LIB", os.path.join(os.path.dirname(_HERE), "libdabgpu.so"))

NB_FFT = 2048
NB_CP = 504
NB_SYM = 2552
NB_NULL = 2656
NB_SYMBOLS = 76
NB_CARRIERS = 1536
NB_FRAME_BITS = 230400
NB_FRAME_SAMPLES = 196608
NB_FIC_BITS = 9216
NB_CIF_BITS = 55296
FRAME_USED_SAMPLES = NB_SYMBOLS * NB_SYM

#: every symbol include/dabgpu.h declares (tests check the library exports all of them)
EXPORTS = [
    "dabgpu_abi_version", "dabgpu_strerror", "dabgpu_get_ofdm_params", "dabgpu_get_dab_params",
    "dabgpu_get_prs_reference", "dabgpu_get_mapper_reference", "dabgpu_create", "dabgpu_destroy",
    "dabgpu_sync", "dabgpu_stream", "dabgpu_ofdm_demod_frames_dev", "dabgpu_ofdm_demod_frames",
    "dabgpu_fft_symbols_dev", "dabgpu_fft_symbols", "dabgpu_fic_decode_dev", "dabgpu_fic_decode",
    "dabgpu_subchannel_bytes", "dabgpu_msc_decode_dev", "dabgpu_msc_decode", "dabgpu_viterbi_dev",
    "dabgpu_viterbi", "dabgpu_set_timing", "dabgpu_last_kernel_ms", "dabgpu_sync_prs_dev", "dabgpu_sync_prs",
    "dabgpu_msc_decode_multi_dev", "dabgpu_dabplus_superframes_dev", "dabgpu_dabplus_superframes",
    "dabgpu_acquire_default_cfg", "dabgpu_acquire_dev", "dabgpu_acquire", "dabgpu_ofdm_demod_acquired_dev",
    "dabgpu_ofdm_set_soft_selection", "dabgpu_soft_selection", "dabgpu_uep_subchannel",
    "dabgpu_host_alloc", "dabgpu_host_free", "dabgpu_decode_frames_dev", "dabgpu_decode_frames",
    "dabgpu_streams_reset", "dabgpu_stream_states", "dabgpu_set_stream_offsets", "dabgpu_ofdm_demod_streams_dev",
    "dabgpu_ofdm_demod_streams", "dabgpu_get_stats", "dabgpu_mean_kernel_ms", "dabgpu_decode_stream_frames",
    "dabgpu_decode_stream_reset", "dabgpu_alloc_frame_buffers", "dabgpu_free_frame_buffers",
    "dabgpu_mover_frames_dev", "dabgpu_pipe_open", "dabgpu_pipe_submit", "dabgpu_pipe_wait", "dabgpu_pipe_reset", "dabgpu_pipe_close",
    "dabgpu_set_stream_loop", "dabgpu_set_loop_gate", "dabgpu_track_default_cfg", "dabgpu_track_start_dev", "dabgpu_ofdm_demod_tracked_dev",
    "dabgpu_ofdm_demod_stream_frame", "dabgpu_ofdm_demod_frames_dd_dev", "dabgpu_test_fail_frame_call",
]

ABI_VERSION = 6
PLACE_PLAIN, PLACE_DOMAINS = 0, 1
PLAIN_ONE_DOMAIN = 7
PLAIN_REASONS = {0: "plain requested", 1: "buffers too small (or too many chunks) for placement", 2: "virtual-memory API refused",
                 3: "no room for the chunks", 4: "the context's domain-aware pair is still alive",
                 5: "larger than the context's reserved address range", 6: "probe launch failed",
                 7: "the placed pair's own check read >= 0.985 (one HBM domain behind the virtual-memory API on this box): "
                    "given back, two plain allocations made instead"}
FLAG_VITERBI_WAVE = 1 << 0
FLAG_VITERBI_LANE = 1 << 1
FLAG_LANE_UNFUSED = 1 << 2
FLAG_TEST_ONE_DOMAIN = 1 << 30          # test hook: the allocator's check of a placed pair reads 1.00


class DabGpuError(RuntimeError):
    def __init__(self, status, what):
        self.status = status
        super().__init__("%s failed: %s (%d)" % (what, strerror(status), status))


class OfdmParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "nb_frame_symbols", "nb_symbol_period", "nb_null_period", "nb_fft", "nb_cyclic_prefix",
        "nb_data_carriers", "freq_carrier_spacing", "nb_frame_samples")]


class DabParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "nb_frame_bits", "nb_symbols", "nb_fic_symbols", "nb_msc_symbols", "nb_sym_bits", "nb_fic_bits",
        "nb_msc_bits", "nb_fibs", "nb_cifs", "nb_fib_bits", "nb_fib_cif_bits", "nb_fibs_per_cif",
        "nb_cif_bits")]


class Cfg(C.Structure):
    _fields_ = [("device", C.c_int32), ("max_frames", C.c_int32), ("transmission_mode", C.c_int32),
                ("flags", C.c_int32), ("ofdm_symbol_runs", C.c_int32), ("reserved", C.c_int32 * 3)]


class Stats(C.Structure):
    _fields_ = [("state", C.c_int32), ("fine_freq_offset", C.c_float), ("coarse_freq_offset", C.c_float),
                ("net_freq_offset", C.c_float), ("signal_average", C.c_float), ("total_frames_read", C.c_int32),
                ("total_frames_desync", C.c_int32), ("last_fine_error", C.c_float), ("tracking", C.c_int32),
                ("last_time_offset", C.c_int32), ("next_frame_start", C.c_double), ("drift", C.c_float),
                ("last_peak_to_mean", C.c_float), ("loop_gated", C.c_int32), ("reserved", C.c_int32)]


STREAM_STATE_DTYPE = np.dtype([("fine_freq_offset", np.float32), ("coarse_freq_offset", np.float32),
                               ("signal_average", np.float32), ("last_fine_error", np.float32),
                               ("total_frames_read", np.int32), ("total_frames_desync", np.int32),
                               ("tracking", np.int32), ("last_time_offset", np.int32),
                               ("next_frame_start", np.float64), ("drift", np.float32), ("last_peak_to_mean", np.float32),
                               ("loop_gated", np.int32), ("dd_branch", np.int32), ("dd_pending", np.int32),
                               ("reserved", np.int32)])     # 64 bytes, device-resident
assert STREAM_STATE_DTYPE.itemsize == 64


class SyncResult(C.Structure):
    _fields_ = [("coarse_carriers", C.c_int32), ("time_offset", C.c_int32), ("peak_to_mean", C.c_float),
                ("coarse_peak_to_mean", C.c_float)]


class AcquireCfg(C.Structure):
    _fields_ = [("thr_null_start", C.c_float), ("thr_null_end", C.c_float), ("min_null_blocks", C.c_int32),
                ("max_coarse_carriers", C.c_int32), ("min_peak_to_mean", C.c_float), ("timing_margin", C.c_int32),
                ("impulse_peak_distance_probability", C.c_float), ("first_path_rel", C.c_float),
                ("level_chunk_blocks", C.c_int32), ("reserved", C.c_int32)]


class TrackCfg(C.Structure):
    _fields_ = [("fine_freq_update_beta", C.c_float), ("signal_update_beta", C.c_float), ("thr_null_start", C.c_float),
                ("min_peak_to_mean", C.c_float), ("impulse_peak_distance_probability", C.c_float),
                ("first_path_rel", C.c_float), ("drift_beta", C.c_float), ("coarse_freq_slow_beta", C.c_float),
                ("timing_margin", C.c_int32), ("max_coarse_carriers", C.c_int32), ("decision_directed", C.c_int32), ("auto_acquire", C.c_int32),
                ("dd_gate", C.c_float), ("reserved", C.c_int32)]


class PlacementReport(C.Structure):
    _fields_ = [("method", C.c_int32), ("fallback_reason", C.c_int32), ("n_chunks", C.c_int32), ("iq_chunks", C.c_int32),
                ("soft_chunks", C.c_int32), ("n_domains", C.c_int32), ("conflicts", C.c_int32), ("runtime_error", C.c_int32),
                ("chunk_bytes", C.c_uint64), ("setup_peak_bytes", C.c_uint64),
                ("classify_ms", C.c_float), ("pair_over_same_domain", C.c_float),
                ("domains", C.c_char * 100), ("iq_map", C.c_char * 72), ("soft_map", C.c_char * 28)]


class FrameResult(C.Structure):
    _fields_ = [("sync", SyncResult), ("flags", C.c_int32), ("reserved", C.c_int32), ("stats", Stats)]


ACQUIRED_FRAME_DTYPE = np.dtype([("start", np.int64), ("freq_offset", np.float32), ("coarse_carriers", np.int32),
                                 ("fine_offset", np.float32), ("peak_to_mean", np.float32),
                                 ("coarse_peak_to_mean", np.float32), ("flags", np.int32)])     # 32 bytes


def soft_selection(subchannels, with_fic=True):
    """[(first_bit, count), ...] covering the FIC and the given sub-channels (dabgpu_soft_selection)."""
    n = len(subchannels)
    arr = (Subchannel * max(n, 1))(*subchannels)
    out = np.zeros((1 + 4 * n, 2), np.int32)
    k = lib().dabgpu_soft_selection(arr, n, int(bool(with_fic)), _p(out), len(out))
    _check(min(k, 0), "dabgpu_soft_selection")
    return [tuple(int(v) for v in r) for r in out[:k]]


def acquire_cfg(**kw):
    """dabgpu_acquire_default_cfg(), then the given fields overridden."""
    c = AcquireCfg()
    lib().dabgpu_acquire_default_cfg(C.byref(c))
    for k, v in kw.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


def track_cfg(**kw):
    """dabgpu_track_default_cfg(), then the given fields overridden."""
    c = TrackCfg()
    lib().dabgpu_track_default_cfg(C.byref(c))
    for k, v in kw.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


class Subchannel(C.Structure):
    _fields_ = [("start_address", C.c_int32), ("length", C.c_int32), ("is_uep", C.c_int32),
                ("eep_type", C.c_int32), ("protection_level", C.c_int32), ("bitrate_kbps", C.c_int32)]


_LIB = None


def load_library(path):
    """Load one build of libdabgpu.so and declare its entry points.  lib() does this for the in-tree library; tools
    that compare builds inside one process (tools/ab_inproc.py) load others and pass them to Context(library=...)."""
    L = C.CDLL(path)
    L.dabgpu_abi_version.argtypes = []
    L.dabgpu_strerror.restype = C.c_char_p
    L.dabgpu_strerror.argtypes = [C.c_int]
    L.dabgpu_get_ofdm_params.argtypes = [C.c_int, C.c_void_p]
    L.dabgpu_get_dab_params.argtypes = [C.c_int, C.c_void_p]
    L.dabgpu_stream.restype = C.c_void_p
    L.dabgpu_stream.argtypes = [C.c_void_p]
    L.dabgpu_destroy.restype = None
    L.dabgpu_destroy.argtypes = [C.c_void_p]
    vp, sz, i = C.c_void_p, C.c_size_t, C.c_int
    L.dabgpu_create.argtypes = [C.POINTER(Cfg), C.POINTER(vp)]
    L.dabgpu_sync.argtypes = [vp]
    L.dabgpu_ofdm_demod_frames_dev.argtypes = [vp, vp, sz, i, vp, vp, vp, vp, vp]
    L.dabgpu_ofdm_demod_frames.argtypes = [vp, vp, sz, i, vp, vp, vp, vp]
    L.dabgpu_ofdm_demod_frames_dd_dev.argtypes = [vp, vp, sz, i, vp, vp, vp, vp]
    L.dabgpu_test_fail_frame_call.argtypes = [vp, i]
    L.dabgpu_fft_symbols_dev.argtypes = [vp, vp, sz, i, vp, vp, vp]
    L.dabgpu_fft_symbols.argtypes = [vp, vp, sz, i, vp, vp]
    L.dabgpu_fic_decode_dev.argtypes = [vp, vp, sz, i, vp, vp, vp]
    L.dabgpu_fic_decode.argtypes = [vp, vp, sz, i, vp, vp]
    L.dabgpu_subchannel_bytes.argtypes = [C.POINTER(Subchannel)]
    L.dabgpu_msc_decode_dev.argtypes = [vp, C.POINTER(Subchannel), vp, sz, i, i, vp, vp, vp, vp]
    L.dabgpu_msc_decode.argtypes = [vp, C.POINTER(Subchannel), vp, sz, i, i, vp, vp, vp]
    L.dabgpu_viterbi_dev.argtypes = [vp, vp, i, vp, i, vp, vp]
    L.dabgpu_viterbi.argtypes = [vp, vp, i, vp, i, vp]
    L.dabgpu_set_timing.argtypes = [vp, i]
    L.dabgpu_dabplus_superframes_dev.argtypes = [vp, vp, sz, i, i, vp, vp, vp]
    L.dabgpu_dabplus_superframes.argtypes = [vp, vp, sz, i, i, vp, vp]
    L.dabgpu_msc_decode_multi_dev.argtypes = [vp, vp, i, vp, sz, i, i, vp, vp, vp, vp]
    L.dabgpu_decode_frames_dev.argtypes = [vp, vp, sz, i, i, vp, vp, vp, i, vp, vp, vp, vp]
    L.dabgpu_decode_frames.argtypes = [vp, vp, sz, i, i, vp, vp, vp, i, vp, vp, vp]
    L.dabgpu_decode_stream_frames.argtypes = [vp, vp, sz, i, vp, vp, vp, i, vp]
    L.dabgpu_decode_stream_reset.argtypes = [vp]
    L.dabgpu_streams_reset.argtypes = [vp, i]
    L.dabgpu_stream_states.restype = C.c_void_p
    L.dabgpu_stream_states.argtypes = [vp]
    L.dabgpu_set_stream_offsets.argtypes = [vp, i, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.dabgpu_ofdm_demod_streams_dev.argtypes = [vp, vp, sz, i, i, C.c_float, vp, vp, vp, vp]
    L.dabgpu_ofdm_demod_streams.argtypes = [vp, vp, sz, i, i, C.c_float, vp, vp, vp]
    L.dabgpu_get_stats.argtypes = [vp, i, C.POINTER(Stats)]
    L.dabgpu_sync_prs_dev.argtypes = [vp, vp, sz, i, vp, i, vp, vp]
    L.dabgpu_sync_prs.argtypes = [vp, vp, sz, i, vp, i, vp]
    L.dabgpu_ofdm_set_soft_selection.argtypes = [vp, vp, i]
    L.dabgpu_soft_selection.argtypes = [vp, i, i, vp, i]
    L.dabgpu_uep_subchannel.argtypes = [i, i, C.POINTER(Subchannel)]
    L.dabgpu_host_alloc.restype = C.c_void_p
    L.dabgpu_host_alloc.argtypes = [sz]
    L.dabgpu_host_free.restype = None
    L.dabgpu_host_free.argtypes = [C.c_void_p]
    L.dabgpu_acquire_default_cfg.restype = None
    L.dabgpu_acquire_default_cfg.argtypes = [C.POINTER(AcquireCfg)]
    L.dabgpu_acquire_dev.argtypes = [vp, vp, sz, i, C.c_int64, C.POINTER(AcquireCfg), i, vp, vp, vp]
    L.dabgpu_acquire.argtypes = [vp, vp, sz, i, C.c_int64, C.POINTER(AcquireCfg), i, vp, vp]
    L.dabgpu_ofdm_demod_acquired_dev.argtypes = [vp, vp, sz, i, i, vp, vp, vp, vp, vp]
    L.dabgpu_last_kernel_ms.argtypes = [vp, i, C.POINTER(C.c_float)]
    L.dabgpu_mean_kernel_ms.argtypes = [vp, i, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    L.dabgpu_alloc_frame_buffers.argtypes = [vp, i, sz, i, C.POINTER(vp), C.POINTER(vp), C.POINTER(PlacementReport)]
    L.dabgpu_free_frame_buffers.argtypes = [vp, vp, vp]
    L.dabgpu_mover_frames_dev.argtypes = [vp, vp, sz, i, vp, i, vp]
    L.dabgpu_pipe_open.argtypes = [vp, i, i, sz]
    L.dabgpu_pipe_submit.argtypes = [vp, vp, i, i, vp, C.c_float, vp, i, vp, vp, vp, vp, C.POINTER(C.c_int64)]
    L.dabgpu_pipe_wait.argtypes = [vp, C.c_int64]
    L.dabgpu_pipe_reset.argtypes = [vp]
    L.dabgpu_pipe_close.argtypes = [vp]
    L.dabgpu_set_stream_loop.argtypes = [vp, C.c_float, C.c_float, i]
    L.dabgpu_set_loop_gate.argtypes = [vp, C.c_float]
    L.dabgpu_track_default_cfg.restype = None
    L.dabgpu_track_default_cfg.argtypes = [C.POINTER(TrackCfg)]
    L.dabgpu_track_start_dev.argtypes = [vp, vp, vp, i, i, C.c_int64, i, vp]
    L.dabgpu_ofdm_demod_tracked_dev.argtypes = [vp, vp, sz, i, C.c_int64, i, C.c_int64, C.POINTER(TrackCfg), vp, vp, vp, vp, vp, vp]
    L.dabgpu_ofdm_demod_stream_frame.argtypes = [vp, i, vp, i, C.POINTER(TrackCfg), vp, vp, C.POINTER(FrameResult)]
    L.dabgpu_get_prs_reference.argtypes = [i, vp, i]
    L.dabgpu_get_mapper_reference.argtypes = [vp, i, i]
    return L


def lib():
    """Load libdabgpu.so (raises if it has not been built -- there is no fallback)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libdabgpu.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "or `make -C sdrplusplus-dab-radio-plugin_amd/csrc` (expected %s)" % LIB_PATH)
        try:
            # share the HIP runtime torch already loaded (same SONAME libamdhip64.so.7)
            import torch  # noqa: F401
        except Exception:
            pass
        _LIB = load_library(LIB_PATH)
    return _LIB


def strerror(status):
    return lib().dabgpu_strerror(C.c_int(status)).decode()


def _check(rc, what):
    if rc != 0:
        raise DabGpuError(rc, what)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ------------------------------------------------------------------ tables (no GPU needed)
def get_ofdm_params(mode=1):
    p = OfdmParams()
    _check(lib().dabgpu_get_ofdm_params(C.c_int(mode), C.byref(p)), "dabgpu_get_ofdm_params")
    return p


def get_dab_params(mode=1):
    p = DabParams()
    _check(lib().dabgpu_get_dab_params(C.c_int(mode), C.byref(p)), "dabgpu_get_dab_params")
    return p


def get_prs_reference(mode=1):
    out = np.zeros(2 * NB_FFT, np.float32)
    _check(lib().dabgpu_get_prs_reference(mode, _p(out), NB_FFT), "dabgpu_get_prs_reference")
    return out.view(np.complex64)


def get_mapper_reference():
    out = np.zeros(NB_CARRIERS, np.int32)
    _check(lib().dabgpu_get_mapper_reference(_p(out), NB_CARRIERS, NB_FFT), "dabgpu_get_mapper_reference")
    return out


def subchannel(start_address, bitrate_kbps, level=3, eep_type=0):
    """EEP subchannel descriptor with the length implied by the profile."""
    if eep_type == 0:
        n = bitrate_kbps // 8
        length = {1: 12 * n, 2: 8 * n, 3: 6 * n, 4: 4 * n}[level]
    else:
        n = bitrate_kbps // 32
        length = {1: 27 * n, 2: 21 * n, 3: 18 * n, 4: 15 * n}[level]
    return Subchannel(start_address, length, 0, eep_type, level, bitrate_kbps)


class PinnedArray:
    """numpy view of page-locked host memory from dabgpu_host_alloc (freed by close() / garbage collection)."""

    def __init__(self, shape, dtype):
        self._n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self._p = lib().dabgpu_host_alloc(self._n)
        if not self._p:
            raise MemoryError("dabgpu_host_alloc(%d)" % self._n)
        buf = (C.c_char * self._n).from_address(self._p)
        self.array = np.frombuffer(buf, dtype=dtype).reshape(shape)

    def close(self):
        if self._p:
            self.array = None
            lib().dabgpu_host_free(self._p)
            self._p = None

    def __del__(self, _finalizing=sys.is_finalizing):   # (bound now: module globals are gone when the interpreter ends)
        try:
            if _finalizing():                    # the HIP runtime may be gone already: the process is about to return it all
                return
            self.close()
        except Exception:
            pass


def device_tensor(torch, address, shape, dtype, device):
    """A torch tensor over device memory this library (or anybody else) owns -- no copy, no ownership: the memory must
    outlive the tensor.  Goes through __cuda_array_interface__, which torch's HIP build honours."""
    typestr = {torch.int8: "|i1", torch.uint8: "|u1", torch.float32: "<f4", torch.int32: "<i4", torch.complex64: "<c8"}[dtype]

    class _Span:
        pass
    span = _Span()
    span.__cuda_array_interface__ = {"shape": tuple(int(x) for x in shape), "typestr": typestr, "data": (int(address), False), "version": 2}
    return torch.as_tensor(span, device=device)


def uep_subchannel(table_index, start_address):
    """UEP subchannel descriptor from the protection-profile table index (FIG 0/1 short form)."""
    sc = Subchannel()
    _check(lib().dabgpu_uep_subchannel(table_index, start_address, C.byref(sc)), "dabgpu_uep_subchannel")
    return sc


# ------------------------------------------------------------------ context
# Contexts still open when the interpreter ends are closed from an atexit handler -- registered when this module is
# imported, i.e. after torch's own, so it runs BEFORE them -- while the HIP runtime is certainly still there; what the
# garbage collector finds after that is left to the operating system.
_LIVE_CONTEXTS = weakref.WeakSet()


def _close_live_contexts():
    for c in list(_LIVE_CONTEXTS):
        try:
            c.close()
        except Exception:
            pass


atexit.register(_close_live_contexts)


def placement_report_dict(rep, requested=None, final_bytes=None):
    """dabgpu_placement_report as a plain dict, field for field (what bench.py prints as config.buffer_placement and every
    rank's row carries): nothing is decided or measured here."""
    d = {"method": "domain-aware pair" if rep.method == 1 else "plain hipMalloc pair",
         "fallback_reason": None if rep.method == 1 else PLAIN_REASONS.get(rep.fallback_reason, str(rep.fallback_reason)),
         "runtime_error": int(rep.runtime_error), "chunks_taken": int(rep.n_chunks), "chunk_bytes": int(rep.chunk_bytes),
         "chunk_domains": rep.domains.decode(), "domains_seen": int(rep.n_domains),
         "iq_chunks": int(rep.iq_chunks), "soft_chunks": int(rep.soft_chunks),
         "iq_chunk_domains": rep.iq_map.decode(), "soft_chunk_domains": rep.soft_map.decode(),
         "soft_bits_written_beside_same_domain_reads_per_mille": int(rep.conflicts),
         "classify_ms": round(float(rep.classify_ms), 2),
         "pair_over_same_domain": round(float(rep.pair_over_same_domain), 3),
         "setup_peak_bytes": int(rep.setup_peak_bytes)}
    if requested is not None:
        d = dict({"requested": requested}, **d)
    if final_bytes:
        d["setup_peak_over_final_footprint"] = round(rep.setup_peak_bytes / final_bytes, 3)
    return d


class Context:
    """Owns a dabgpu_ctx.  Host-array methods copy in/out and synchronise; *_dev methods take
    raw device addresses (e.g. torch.Tensor.data_ptr()) and a stream handle and only enqueue."""

    def __init__(self, device=0, max_frames=64, flags=0, ofdm_symbol_runs=0, library=None):
        self._h = C.c_void_p()
        self._device = int(device)
        self._lib = library if library is not None else lib()
        cfg = Cfg(device, max_frames, 1, flags, ofdm_symbol_runs)
        _check(self._lib.dabgpu_create(C.byref(cfg), C.byref(self._h)), "dabgpu_create")
        _LIVE_CONTEXTS.add(self)

    def close(self):
        if self._h:
            self._lib.dabgpu_destroy(self._h)
            self._h = C.c_void_p()
        _LIVE_CONTEXTS.discard(self)

    def __del__(self, _finalizing=sys.is_finalizing):
        try:
            if _finalizing():                    # (see _close_live_contexts: contexts are closed before the interpreter goes)
                return
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def stream(self):
        return self._lib.dabgpu_stream(self._h)

    def sync(self):
        _check(self._lib.dabgpu_sync(self._h), "dabgpu_sync")

    def set_timing(self, on):
        _check(self._lib.dabgpu_set_timing(self._h, 1 if on else 0), "dabgpu_set_timing")

    def last_kernel_ms(self, which):
        ms = C.c_float(0)
        _check(self._lib.dabgpu_last_kernel_ms(self._h, which, C.byref(ms)), "dabgpu_last_kernel_ms")
        return ms.value

    def mean_kernel_ms(self, which):
        """(mean ms, launches) of kernel family `which` since set_timing(True) (at most the last 32 launches)."""
        ms, n = C.c_float(0), C.c_int(0)
        _check(self._lib.dabgpu_mean_kernel_ms(self._h, which, C.byref(ms), C.byref(n)), "dabgpu_mean_kernel_ms")
        return ms.value, n.value

    # ---- closed-loop stream call
    def alloc_frame_buffers(self, n_frames, frame_stride=NB_FRAME_SAMPLES, placement=PLACE_PLAIN):
        """Device buffers for [n_frames][frame_stride] cf32 samples and [n_frames][230400] soft bits
        (dabgpu_alloc_frame_buffers): two hipMallocs, or (PLACE_DOMAINS) placed by HBM domain inside 1.5 x their size.
        -> (d_iq, d_soft, PlacementReport): raw device addresses; release with free_frame_buffers."""
        d_iq, d_soft = C.c_void_p(), C.c_void_p()
        rep = PlacementReport()
        _check(self._lib.dabgpu_alloc_frame_buffers(self._h, n_frames, frame_stride, placement, C.byref(d_iq), C.byref(d_soft),
                                                    C.byref(rep)), "dabgpu_alloc_frame_buffers")
        return d_iq.value, d_soft.value, rep

    def test_fail_frame_call(self, nth):
        """Test hook: the nth one-frame call from now on reports DABGPU_ERR_HIP before any launch (0 disarms)."""
        _check(self._lib.dabgpu_test_fail_frame_call(self._h, nth), "dabgpu_test_fail_frame_call")

    def mover_frames_dev(self, d_iq, frame_stride, n_frames, d_soft, with_prefixes=False, stream=None):
        """The front end's loads and stores without its arithmetic (dabgpu_mover_frames_dev); overwrites d_soft."""
        _check(self._lib.dabgpu_mover_frames_dev(self._h, d_iq, frame_stride, n_frames, d_soft, 1 if with_prefixes else 0, stream),
               "dabgpu_mover_frames_dev")

    # ---- host-fed ring
    def pipe_open(self, slots, max_frames, frame_stride=FRAME_USED_SAMPLES):
        _check(self._lib.dabgpu_pipe_open(self._h, slots, max_frames, frame_stride), "dabgpu_pipe_open")

    def pipe_submit(self, iq, n_streams, frames_per_stream, freq_offset, scs, soft, fib, crc_ok, outs, beta=0.9):
        """Enqueue one batch (dabgpu_pipe_submit).  Every array argument is a numpy array (or None for freq_offset / soft)
        that the CALLER keeps alive and untouched until pipe_wait(ticket); page-locked ones (PinnedArray.array) move at the
        link rate.  -> ticket"""
        n = len(scs)
        arr = (Subchannel * max(n, 1))(*scs)
        ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in outs]) if n else None
        t = C.c_int64(-1)
        _check(self._lib.dabgpu_pipe_submit(self._h, iq.ctypes.data, n_streams, frames_per_stream,
                                            None if freq_offset is None else freq_offset.ctypes.data, beta, arr, n,
                                            None if soft is None else soft.ctypes.data, fib.ctypes.data, crc_ok.ctypes.data, ptrs,
                                            C.byref(t)), "dabgpu_pipe_submit")
        return t.value

    def pipe_wait(self, ticket):
        _check(self._lib.dabgpu_pipe_wait(self._h, ticket), "dabgpu_pipe_wait")

    def pipe_reset(self):
        _check(self._lib.dabgpu_pipe_reset(self._h), "dabgpu_pipe_reset")

    def pipe_close(self):
        _check(self._lib.dabgpu_pipe_close(self._h), "dabgpu_pipe_close")

    def free_frame_buffers(self, d_iq, d_soft):
        _check(self._lib.dabgpu_free_frame_buffers(self._h, d_iq, d_soft), "dabgpu_free_frame_buffers")

    def streams_reset(self, n_streams):
        _check(self._lib.dabgpu_streams_reset(self._h, n_streams), "dabgpu_streams_reset")

    @property
    def stream_states_ptr(self):
        return self._lib.dabgpu_stream_states(self._h)

    def set_stream_offsets(self, stream, fine=None, coarse=None):
        f = None if fine is None else C.byref(C.c_float(fine))
        c = None if coarse is None else C.byref(C.c_float(coarse))
        _check(self._lib.dabgpu_set_stream_offsets(self._h, stream, f, c), "dabgpu_set_stream_offsets")

    def get_stats(self, stream):
        st = Stats()
        _check(self._lib.dabgpu_get_stats(self._h, stream, C.byref(st)), "dabgpu_get_stats")
        return st

    def set_stream_loop(self, signal_update_beta=0.95, thr_null_start=0.35, decision_directed=False):
        _check(self._lib.dabgpu_set_stream_loop(self._h, signal_update_beta, thr_null_start, int(bool(decision_directed))),
               "dabgpu_set_stream_loop")

    def stream_states_host(self, n_streams):
        """The first n_streams stream states as a numpy record array (STREAM_STATE_DTYPE), read back after waiting for the
        context stream.  The caller syncs whatever other stream the last stream call ran on."""
        import torch
        self.sync()
        t = device_tensor(torch, self.stream_states_ptr, (n_streams * 64,), torch.uint8, torch.device("cuda", self._device))
        torch.cuda.synchronize()
        return t.cpu().numpy().view(STREAM_STATE_DTYPE).copy()

    def set_loop_gate(self, dd_gate):
        _check(self._lib.dabgpu_set_loop_gate(self._h, dd_gate), "dabgpu_set_loop_gate")

    def track_start_dev(self, d_frames, d_counts, n_streams, max_frames, advance, stream=None, only_lost=False):
        _check(self._lib.dabgpu_track_start_dev(self._h, d_frames, d_counts, n_streams, max_frames, advance, int(bool(only_lost)),
                                            stream), "dabgpu_track_start_dev")

    def ofdm_demod_tracked_dev(self, d_iq, stream_stride, n_streams, n_samples, max_frames, advance, d_soft, d_frames, d_counts,
                               cfg=None, d_cyc=None, d_dqpsk=None, stream=None):
        _check(self._lib.dabgpu_ofdm_demod_tracked_dev(self._h, d_iq, stream_stride, n_streams, n_samples, max_frames, advance,
                                                   None if cfg is None else C.byref(cfg), d_soft, d_cyc, d_dqpsk, d_frames,
                                                   d_counts, stream), "dabgpu_ofdm_demod_tracked_dev")

    def ofdm_demod_stream_frame(self, iq, stream=0, acquiring=False, cfg=None, want_dqpsk=False):
        """One frame (76*2552 cf32, host) of stream `stream` in one call -> soft [230400], FrameResult, dqpsk or None."""
        iq = np.ascontiguousarray(iq, np.complex64).reshape(-1)
        assert iq.size >= FRAME_USED_SAMPLES
        soft = np.zeros(NB_FRAME_BITS, np.int8)
        dq = np.zeros((NB_SYMBOLS - 1, NB_CARRIERS), np.complex64) if want_dqpsk else None
        res = FrameResult()
        _check(self._lib.dabgpu_ofdm_demod_stream_frame(self._h, stream, _p(iq), int(bool(acquiring)),
                                                    None if cfg is None else C.byref(cfg), _p(soft), _p(dq), C.byref(res)),
               "dabgpu_ofdm_demod_stream_frame")
        return soft, res, dq

    def ofdm_demod_streams(self, iq, n_streams, beta=0.9, want_cyc=False, soft=None):
        """iq: complex64 [n_streams*frames_per_stream][>=76*2552]; uses and updates the context's stream states.
        `soft` may be an existing array (selected ranges only are overwritten when a selection is active)."""
        iq = np.ascontiguousarray(iq, np.complex64)
        n_frames, stride = iq.shape
        if soft is None:
            soft = np.zeros((n_frames, NB_FRAME_BITS), np.int8)
        cyc = np.zeros((n_frames, NB_SYMBOLS), np.complex64) if want_cyc else None
        _check(self._lib.dabgpu_ofdm_demod_streams(self._h, _p(iq), stride, n_streams, n_frames // n_streams, beta, _p(soft),
                                               _p(cyc), None), "dabgpu_ofdm_demod_streams")
        return soft, cyc

    def ofdm_demod_streams_dev(self, d_iq, frame_stride, n_streams, frames_per_stream, beta, d_soft, d_cyc=None,
                               d_dqpsk=None, stream=None):
        _check(self._lib.dabgpu_ofdm_demod_streams_dev(self._h, d_iq, frame_stride, n_streams, frames_per_stream, beta, d_soft,
                                                   d_cyc, d_dqpsk, stream), "dabgpu_ofdm_demod_streams_dev")

    def decode_frames(self, soft, n_streams, scs, history_in=None, want_history=False):
        """Host arrays: FIC + sub-channels `scs` of soft [n_streams*frames_per_stream][230400] in one call
        (dabgpu_decode_frames).  -> fib, crc_ok, [out_i], [history_out_i] (or None)"""
        soft = np.ascontiguousarray(soft, np.int8)
        n_frames, stride = soft.shape
        fps = n_frames // n_streams
        n = len(scs)
        arr = (Subchannel * max(n, 1))(*scs)
        fib = np.zeros((n_frames, 12, 32), np.uint8)
        ok = np.zeros((n_frames, 12), np.uint8)
        outs = []
        for sc in scs:
            nb = self._lib.dabgpu_subchannel_bytes(C.byref(sc))
            _check(min(nb, 0), "dabgpu_subchannel_bytes")
            outs.append(np.zeros((n_streams, fps * 4, nb), np.uint8))
        his = [None if history_in is None or history_in[k] is None else np.ascontiguousarray(history_in[k], np.int8)
               for k in range(n)]
        hos = [np.zeros((n_streams, 15, sc.length * 64), np.int8) if want_history else None for sc in scs]

        def ptrs(lst):
            if n == 0:
                return None
            return (C.c_void_p * n)(*[None if a is None else a.ctypes.data for a in lst])
        _check(self._lib.dabgpu_decode_frames(self._h, _p(soft), stride, n_streams, fps, _p(fib), _p(ok), arr, n, ptrs(his),
                                          ptrs(hos), ptrs(outs)), "dabgpu_decode_frames")
        return fib, ok, outs, (hos if want_history else None)

    def decode_stream_frames(self, soft, scs):
        """Consecutive frames of one stream; the de-interleaver state stays in the context.  -> fib, crc_ok, [out_i]"""
        soft = np.ascontiguousarray(soft, np.int8)
        n_frames, stride = soft.shape
        n = len(scs)
        arr = (Subchannel * max(n, 1))(*scs)
        fib = np.zeros((n_frames, 12, 32), np.uint8)
        ok = np.zeros((n_frames, 12), np.uint8)
        outs = []
        for sc in scs:
            nb = self._lib.dabgpu_subchannel_bytes(C.byref(sc))
            _check(min(nb, 0), "dabgpu_subchannel_bytes")
            outs.append(np.zeros((1, n_frames * 4, nb), np.uint8))
        ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in outs]) if n else None
        _check(self._lib.dabgpu_decode_stream_frames(self._h, _p(soft), stride, n_frames, _p(fib), _p(ok), arr, n, ptrs),
               "dabgpu_decode_stream_frames")
        return fib, ok, outs

    def decode_stream_reset(self):
        _check(self._lib.dabgpu_decode_stream_reset(self._h), "dabgpu_decode_stream_reset")

    # ---- host arrays
    def ofdm_demod_frames(self, iq, freq_offset=None, want_cyc=False, want_dqpsk=False, soft=None):
        """iq: complex64 [n_frames][>=76*2552], row f starting at the first PRS sample."""
        iq = np.ascontiguousarray(iq, np.complex64)
        n_frames, stride = iq.shape
        if soft is None:
            soft = np.zeros((n_frames, NB_FRAME_BITS), np.int8)
        fo = None if freq_offset is None else np.ascontiguousarray(freq_offset, np.float32)
        cyc = np.zeros((n_frames, NB_SYMBOLS), np.complex64) if want_cyc else None
        dq = np.zeros((n_frames, NB_SYMBOLS - 1, NB_CARRIERS), np.complex64) if want_dqpsk else None
        _check(self._lib.dabgpu_ofdm_demod_frames(self._h, _p(iq), stride, n_frames, _p(fo), _p(soft), _p(cyc), _p(dq)),
               "dabgpu_ofdm_demod_frames")
        return soft, cyc, dq

    def fft_symbols(self, iq, freq_offset=None):
        iq = np.ascontiguousarray(iq, np.complex64)
        n_frames, stride = iq.shape
        fo = None if freq_offset is None else np.ascontiguousarray(freq_offset, np.float32)
        out = np.zeros((n_frames, NB_SYMBOLS, NB_FFT), np.complex64)
        _check(self._lib.dabgpu_fft_symbols(self._h, _p(iq), stride, n_frames, _p(fo), _p(out)), "dabgpu_fft_symbols")
        return out

    def sync_prs(self, iq, freq_offset=None, max_coarse=200):
        """iq: complex64 [n][>=2552], row f starting at the candidate first sample of the PRS cyclic prefix.
        -> structured array with coarse_carriers, time_offset, peak_to_mean, coarse_peak_to_mean."""
        iq = np.ascontiguousarray(iq, np.complex64)
        n, stride = iq.shape
        fo = None if freq_offset is None else np.ascontiguousarray(freq_offset, np.float32)
        out = np.zeros(n, dtype=[("coarse_carriers", np.int32), ("time_offset", np.int32),
                                 ("peak_to_mean", np.float32), ("coarse_peak_to_mean", np.float32)])
        _check(self._lib.dabgpu_sync_prs(self._h, _p(iq), stride, n, _p(fo), max_coarse, _p(out)), "dabgpu_sync_prs")
        return out

    def set_soft_selection(self, ranges):
        """ranges: iterable of (first_bit, count) in frame-bit coordinates, or None / empty for whole frames."""
        arr = np.array(list(ranges) if ranges is not None else [], np.int32).reshape(-1, 2)
        _check(self._lib.dabgpu_ofdm_set_soft_selection(self._h, _p(arr) if len(arr) else None, len(arr)),
               "dabgpu_ofdm_set_soft_selection")

    def acquire(self, iq, max_frames, cfg=None):
        """iq: complex64 [n_streams][n_samples] unaligned captures -> (frames [n_streams][max_frames] structured
        array (ACQUIRED_FRAME_DTYPE), counts [n_streams])."""
        iq = np.ascontiguousarray(iq, np.complex64)
        n_streams, n_samples = iq.shape
        out = np.zeros((n_streams, max_frames), ACQUIRED_FRAME_DTYPE)
        counts = np.zeros(n_streams, np.int32)
        _check(self._lib.dabgpu_acquire(self._h, _p(iq), n_samples, n_streams, n_samples,
                                    None if cfg is None else C.byref(cfg), max_frames, _p(out), _p(counts)), "dabgpu_acquire")
        return out, counts

    def acquire_dev(self, d_iq, stream_stride, n_streams, n_samples, max_frames, d_out, d_counts, cfg=None, stream=None):
        _check(self._lib.dabgpu_acquire_dev(self._h, d_iq, stream_stride, n_streams, n_samples,
                                        None if cfg is None else C.byref(cfg), max_frames, d_out, d_counts, stream),
               "dabgpu_acquire_dev")

    def ofdm_demod_acquired_dev(self, d_iq, stream_stride, n_streams, max_frames, d_frames, d_soft, d_cyc=None,
                                d_dqpsk=None, stream=None):
        _check(self._lib.dabgpu_ofdm_demod_acquired_dev(self._h, d_iq, stream_stride, n_streams, max_frames, d_frames, d_soft,
                                                    d_cyc, d_dqpsk, stream), "dabgpu_ofdm_demod_acquired_dev")

    def dabplus_superframes(self, sfs, bitrate_kbps):
        """sfs: uint8 [n][>=15*bitrate] aligned super-frames -> (data [n][110*s], status structured array)."""
        sfs = np.ascontiguousarray(sfs, np.uint8)
        n, stride = sfs.shape
        s = bitrate_kbps // 8
        out = np.zeros((n, 110 * s), np.uint8)
        st = np.zeros(n, dtype=[("firecode_ok", np.int32), ("rs_corrected", np.int32), ("rs_uncorrectable", np.int32),
                                ("num_aus", np.int32), ("au_crc_mask", np.int32), ("au_start", np.int32, (8,)),
                                ("reserved", np.int32, (3,))])
        _check(self._lib.dabgpu_dabplus_superframes(self._h, _p(sfs), stride, n, bitrate_kbps, _p(out), _p(st)),
               "dabgpu_dabplus_superframes")
        return out, st

    def fic_decode(self, soft):
        """soft: int8 [n_frames][>=9216]."""
        soft = np.ascontiguousarray(soft, np.int8)
        n_frames, stride = soft.shape
        fib = np.zeros((n_frames, 12, 32), np.uint8)
        ok = np.zeros((n_frames, 12), np.uint8)
        _check(self._lib.dabgpu_fic_decode(self._h, _p(soft), stride, n_frames, _p(fib), _p(ok)), "dabgpu_fic_decode")
        return fib, ok

    def msc_decode(self, sc, soft, n_streams, history_in=None, want_history=False):
        """soft: int8 [n_streams*frames_per_stream][230400]."""
        soft = np.ascontiguousarray(soft, np.int8)
        n_frames, stride = soft.shape
        fps = n_frames // n_streams
        nbytes = self._lib.dabgpu_subchannel_bytes(C.byref(sc))
        _check(min(nbytes, 0), "dabgpu_subchannel_bytes")
        out = np.zeros((n_streams, fps * 4, nbytes), np.uint8)
        hi = None if history_in is None else np.ascontiguousarray(history_in, np.int8)
        ho = np.zeros((n_streams, 15, sc.length * 64), np.int8) if want_history else None
        _check(self._lib.dabgpu_msc_decode(self._h, C.byref(sc), _p(soft), stride, n_streams, fps, _p(hi), _p(ho), _p(out)),
               "dabgpu_msc_decode")
        return out, ho

    def viterbi(self, punct, mask):
        """punct: int8 [n_codewords][n_punct]; mask: uint8 [4*nsteps] -> bytes [n][(nsteps-6)/8]."""
        punct = np.ascontiguousarray(punct, np.int8)
        mask = np.ascontiguousarray(mask, np.uint8)
        nsteps = mask.size // 4
        n = punct.shape[0]
        out = np.zeros((n, (nsteps - 6) // 8), np.uint8)
        _check(self._lib.dabgpu_viterbi(self._h, _p(punct), n, _p(mask), nsteps, _p(out)), "dabgpu_viterbi")
        return out

    # ---- device pointers (ints), enqueue only
    def ofdm_demod_frames_dev(self, d_iq, frame_stride, n_frames, d_freq_offset, d_soft, d_cyc=None, d_dqpsk=None,
                              stream=None):
        _check(self._lib.dabgpu_ofdm_demod_frames_dev(self._h, d_iq, frame_stride, n_frames, d_freq_offset, d_soft,
                                                  d_cyc, d_dqpsk, stream), "dabgpu_ofdm_demod_frames_dev")

    def ofdm_demod_frames_dd_dev(self, d_iq, frame_stride, n_frames, d_freq_offset, d_soft, d_dd4, stream=None):
        _check(self._lib.dabgpu_ofdm_demod_frames_dd_dev(self._h, d_iq, frame_stride, n_frames, d_freq_offset, d_soft, d_dd4,
                                                     stream), "dabgpu_ofdm_demod_frames_dd_dev")

    def sync_prs_dev(self, d_iq, frame_stride, n_frames, d_freq_offset, max_coarse, d_out, stream=None):
        """d_out: [n_frames] dabgpu_sync_result (4 x 32 bit: coarse_carriers, time_offset, peak_to_mean, coarse ptm)."""
        _check(self._lib.dabgpu_sync_prs_dev(self._h, d_iq, frame_stride, n_frames, d_freq_offset, max_coarse, d_out, stream),
               "dabgpu_sync_prs_dev")

    def fft_symbols_dev(self, d_iq, frame_stride, n_frames, d_freq_offset, d_spectra, stream=None):
        _check(self._lib.dabgpu_fft_symbols_dev(self._h, d_iq, frame_stride, n_frames, d_freq_offset, d_spectra, stream),
               "dabgpu_fft_symbols_dev")

    def fic_decode_dev(self, d_soft, soft_stride, n_frames, d_fib, d_crc_ok, stream=None):
        _check(self._lib.dabgpu_fic_decode_dev(self._h, d_soft, soft_stride, n_frames, d_fib, d_crc_ok, stream),
               "dabgpu_fic_decode_dev")

    def msc_decode_multi_dev(self, scs, d_soft, soft_stride, n_streams, frames_per_stream, d_hist_in, d_hist_out, d_out,
                             stream=None):
        """scs: list of Subchannel; d_hist_in/out, d_out: lists of device addresses (or None)."""
        n = len(scs)
        arr = (Subchannel * n)(*scs)
        def ptrs(lst):
            if lst is None:
                return None
            return (C.c_void_p * n)(*[C.c_void_p(x) if x else None for x in lst])
        hi, ho, out = ptrs(d_hist_in), ptrs(d_hist_out), ptrs(d_out)
        _check(self._lib.dabgpu_msc_decode_multi_dev(self._h, arr, n, d_soft, soft_stride, n_streams, frames_per_stream,
                                                 hi, ho, out, stream), "dabgpu_msc_decode_multi_dev")

    def decode_frames_dev(self, d_soft, soft_stride, n_streams, frames_per_stream, d_fib, d_crc_ok, scs, d_hist_in, d_hist_out,
                          d_out, stream=None):
        """FIC + the sub-channels `scs` of every frame in one call (lists of device addresses as msc_decode_multi_dev)."""
        n = len(scs)
        arr = (Subchannel * max(n, 1))(*scs)
        def ptrs(lst):
            if lst is None or n == 0:
                return None
            return (C.c_void_p * n)(*[C.c_void_p(x) if x else None for x in lst])
        _check(self._lib.dabgpu_decode_frames_dev(self._h, d_soft, soft_stride, n_streams, frames_per_stream, d_fib, d_crc_ok,
                                              arr, n, ptrs(d_hist_in), ptrs(d_hist_out), ptrs(d_out), stream),
               "dabgpu_decode_frames_dev")

    def msc_decode_dev(self, sc, d_soft, soft_stride, n_streams, frames_per_stream, d_hist_in, d_hist_out, d_out,
                       stream=None):
        _check(self._lib.dabgpu_msc_decode_dev(self._h, C.byref(sc), d_soft, soft_stride, n_streams, frames_per_stream,
                                           d_hist_in, d_hist_out, d_out, stream), "dabgpu_msc_decode_dev")
