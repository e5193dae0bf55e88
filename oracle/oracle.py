"""ctypes binding of the CPU oracle (oracle/dab_oracle.c).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product path (sdrplusplus-dab-radio-plugin_amd/)
never imports this module.  PARITY UNPINNED: see dab_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

NB_FFT = 2048
NB_CP = 504
NB_SYM = 2552
NB_NULL = 2656
NB_SYMBOLS = 76
NB_CARRIERS = 1536
NB_SYM_BITS = 3072
NB_FRAME_BITS = 230400
NB_FRAME_SAMPLES = 196608
NB_FIC_BITS = 9216
NB_CIF_BITS = 55296


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, f) for f in ("dab_oracle.c", "dab_oracle.h", "oracle_bench.c", "dabplus_oracle.c")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        # ORACLE_LIB lets the CPU suite run against the ASan/UBSan build (make -C oracle liboracle_asan.so)
        _LIB = C.CDLL(os.environ.get("ORACLE_LIB") or build())
        L = _LIB
        L.oracle_carrier_bin.restype = C.c_int
        L.oracle_fic_puncture_mask.restype = C.c_int
        L.oracle_eep_puncture_mask.restype = C.c_int
        L.oracle_crc16.restype = C.c_uint16
        L.oracle_bench_frames.restype = C.c_double
        L.oracle_firecode.restype = C.c_uint16
        L.oracle_rs_decode.restype = C.c_int
        L.oracle_bench_frames_timed.restype = C.c_double
        L.oracle_bench_ofdm_only_timed.restype = C.c_double
        L.oracle_bench_pipeline_timed.restype = C.c_double
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def mapper():
    out = np.zeros(NB_CARRIERS, np.int32)
    lib().oracle_get_mapper(_p(out))
    return out


def carrier_bins():
    return np.array([lib().oracle_carrier_bin(i) for i in range(NB_CARRIERS)], np.int32)


def prs():
    out = np.zeros(2 * NB_FFT, np.float32)
    lib().oracle_get_prs(_p(out))
    return out.view(np.complex64)


def puncture_vector(pi):
    out = np.zeros(32, np.uint8)
    lib().oracle_puncture_vector(C.c_int(pi), _p(out))
    return out


def fic_puncture_mask():
    m = np.zeros(3096, np.uint8)
    kept = lib().oracle_fic_puncture_mask(_p(m))
    return m, kept


def eep_puncture_mask(option, level, bitrate):
    m = np.zeros(4 * (bitrate * 24 + 6), np.uint8)
    nsteps = C.c_int(0)
    cu = C.c_int(0)
    kept = lib().oracle_eep_puncture_mask(C.c_int(option), C.c_int(level), C.c_int(bitrate), _p(m),
                                          C.byref(nsteps), C.byref(cu))
    if kept < 0:
        raise ValueError("invalid EEP profile")
    return m, kept, nsteps.value, cu.value


def uep_profile(index):
    br, lv, cu = C.c_int(0), C.c_int(0), C.c_int(0)
    if lib().oracle_uep_profile(C.c_int(index), C.byref(br), C.byref(lv), C.byref(cu)) != 0:
        raise ValueError("invalid UEP table index")
    return br.value, lv.value, cu.value


def uep_puncture_mask(index):
    """-> (mask, transmitted bits without padding, nsteps, size_cu)."""
    br, _, _ = uep_profile(index)
    m = np.zeros(4 * (br * 24 + 6), np.uint8)
    nsteps, kept, cu = C.c_int(0), C.c_int(0), C.c_int(0)
    if lib().oracle_uep_puncture_mask(C.c_int(index), _p(m), C.byref(nsteps), C.byref(kept), C.byref(cu)) != 0:
        raise ValueError("invalid UEP profile")
    return m, kept.value, nsteps.value, cu.value


def prbs(n):
    out = np.zeros(n, np.uint8)
    lib().oracle_prbs(_p(out), C.c_int(n))
    return out


def crc16(data):
    a = np.ascontiguousarray(data, np.uint8)
    return int(lib().oracle_crc16(_p(a), C.c_int(a.size)))


def conv_encode(bits):
    b = np.ascontiguousarray(bits, np.uint8)
    out = np.zeros(4 * (b.size + 6), np.uint8)
    lib().oracle_conv_encode(_p(b), C.c_int(b.size), _p(out))
    return out


def viterbi(mother_soft):
    s = np.ascontiguousarray(mother_soft, np.int8)
    nsteps = s.size // 4
    out = np.zeros(nsteps - 6, np.uint8)
    lib().oracle_viterbi(_p(s), C.c_int(nsteps), _p(out))
    return out


def depuncture(punct, mask):
    p = np.ascontiguousarray(punct, np.int8)
    m = np.ascontiguousarray(mask, np.uint8)
    out = np.zeros(m.size, np.int8)
    lib().oracle_depuncture(_p(p), _p(m), C.c_int(m.size), _p(out))
    return out


def fic_decode(soft):
    s = np.ascontiguousarray(soft[:NB_FIC_BITS], np.int8)
    fib = np.zeros(384, np.uint8)
    ok = np.zeros(12, np.uint8)
    lib().oracle_fic_decode(_p(s), _p(fib), _p(ok))
    return fib.reshape(12, 32), ok


def time_deinterleave(cifs16):
    """cifs16: [16][nbits] int8, row k = CIF (t-15+k)."""
    a = np.ascontiguousarray(cifs16, np.int8)
    nbits = a.shape[1]
    ptrs = (C.c_void_p * 16)(*[a[k].ctypes.data for k in range(16)])
    out = np.zeros(nbits, np.int8)
    lib().oracle_time_deinterleave(ptrs, C.c_int(nbits), _p(out))
    return out


def msc_decode_lf(deint, mask, nsteps):
    d = np.ascontiguousarray(deint, np.int8)
    m = np.ascontiguousarray(mask, np.uint8)
    out = np.zeros((nsteps - 6) // 8, np.uint8)
    lib().oracle_msc_decode_lf(_p(d), _p(m), C.c_int(nsteps), _p(out))
    return out


def fft2048(x):
    a = np.ascontiguousarray(x, np.complex64)
    out = np.zeros(NB_FFT, np.complex64)
    lib().oracle_fft2048(_p(a), _p(out))
    return out


def ofdm_demod_frame(iq, freq_offset=0.0, want_spectra=False, want_cyc=False, want_dqpsk=False):
    """iq: complex64[76*2552] starting at the first PRS sample."""
    a = np.ascontiguousarray(iq, np.complex64)
    assert a.size == NB_SYMBOLS * NB_SYM
    soft = np.zeros(NB_FRAME_BITS, np.int8)
    spectra = np.zeros((NB_SYMBOLS, NB_FFT), np.complex64) if want_spectra else None
    cyc = np.zeros(NB_SYMBOLS, np.complex64) if want_cyc else None
    dq = np.zeros((NB_SYMBOLS - 1, NB_CARRIERS), np.complex64) if want_dqpsk else None
    lib().oracle_ofdm_demod_frame(_p(a), C.c_float(freq_offset), _p(soft),
                                  _p(spectra) if want_spectra else None,
                                  _p(cyc) if want_cyc else None,
                                  _p(dq) if want_dqpsk else None)
    return soft, spectra, cyc, dq


def ofdm_demod_frame_dd(iq, freq_offset=0.0):
    """-> (soft, dd4): dd4 complex64[76], the decision-directed frequency-error sums (entry 0: the PRS's cyclic-prefix
    correlation)."""
    a = np.ascontiguousarray(iq, np.complex64)
    assert a.size == NB_SYMBOLS * NB_SYM
    soft = np.zeros(NB_FRAME_BITS, np.int8)
    dd4 = np.zeros(NB_SYMBOLS, np.complex64)
    lib().oracle_ofdm_demod_frame_dd(_p(a), C.c_float(freq_offset), _p(soft), None, None, None, _p(dd4))
    return soft, dd4


def dd_error(dd4):
    """Residual frequency offset (cycles/sample) from dd4 rows of any shape [..., 76]: the fourth-power estimate
    angle(-sum over l >= 1) / (4 2 pi 2552), whose branch (it repeats every 1 / (4 2552)) is picked by the mean angle of the
    PRS cyclic-prefix correlations in entry 0, / (2 pi 2048)."""
    a = np.asarray(dd4, np.complex128)
    s = a[..., 1:].sum()
    e_dd = np.float32(np.arctan2(-s.imag, -s.real) / (4.0 * 2.0 * np.pi * 2552.0))
    e_cp = np.float32(np.float32(np.angle(a[..., 0]).mean()) * np.float32(1.0 / (6.283185307179586 * 2048.0)))
    k = np.rint(np.float32(e_cp - e_dd) * np.float32(4.0 * 2552.0))
    return np.float32(e_dd + np.float32(k) * np.float32(1.0 / (4.0 * 2552.0)))


DD_NO_BRANCH = 0x7fffffff


def dd_loop_error(dd4, state, gate=2.5, terms_per_frame=19200):
    """The decision-directed fine-frequency error of a call WITH its two gates (TEST INFRASTRUCTURE; restates
    dabk::dd_loop_error, csrc/kernels.hpp -- an estimator of this library's own: the reference shows only that a fine loop
    exists, /root/reference/src/render_radio_block.cpp:202, :216).  dd4: rows [frames][76] (entry 0 = PRS cyclic-prefix
    correlation, entries 1.. = fourth-power sums); state: dict with total_frames_read and (optionally) loop_gated,
    dd_branch, dd_pending.  Returns (err float32, new {loop_gated, dd_branch, dd_pending}).
      quality  |sum|^2 < gate^2 * n_terms: the cyclic-prefix estimate of the PRSs alone
      branch   previous branch 0 and now +-1, not seen on the call before: held at 0 once"""
    a = np.asarray(dd4, np.complex128)
    frames = a.reshape(-1, a.shape[-1]).shape[0]
    s = a[..., 1:].sum()
    e_cp = np.float32(np.float32(np.angle(a[..., 0]).sum() / frames) * np.float32(1.0 / (6.283185307179586 * 2048.0)))
    gated = int(state.get("loop_gated", 0))
    branch = int(state.get("dd_branch", 0))
    pending = int(state.get("dd_pending", 0))
    n_terms = float(frames) * float(terms_per_frame)
    g = float(np.float32(gate))
    if not (n_terms > 0.0) or (s.real * s.real + s.imag * s.imag) < g * g * n_terms:
        k = int(np.rint(np.float32(e_cp * np.float32(4.0 * 2552.0))))
        return e_cp, {"loop_gated": gated + 1, "dd_branch": k, "dd_pending": DD_NO_BRANCH}
    e_dd = np.float32(np.arctan2(-s.imag, -s.real) / (4.0 * 6.283185307179586 * 2552.0))
    k = int(np.rint(np.float32(np.float32(e_cp - e_dd) * np.float32(4.0 * 2552.0))))
    first = int(state.get("total_frames_read", 0)) == 0
    if (not first) and branch == 0 and k in (1, -1) and pending != k:
        pending, gated, k = k, gated + 1, 0
    else:
        pending = DD_NO_BRANCH
    err = np.float32(e_dd + np.float32(np.float32(k) * np.float32(1.0 / (4.0 * 2552.0))))
    return err, {"loop_gated": gated, "dd_branch": k, "dd_pending": pending}


def stream_update(state, cyc, iq_last_frame, beta, thr_null_start=0.35, signal_beta=0.95, dd=False, dd_gate=2.5, terms_per_frame=19200):
    """Restatement of the fine-frequency loop and counters of the stream call (TEST INFRASTRUCTURE; parity unpinned:
    the loop runs inside the absent DAB-Radio OFDM_Demod, its existence and knobs are visible at
    /root/reference/src/render_radio_block.cpp:202-207, :216).  state: dict with fine_freq_offset,
    coarse_freq_offset, signal_average, total_frames_read, total_frames_desync (decision-directed: + loop_gated, dd_branch,
    dd_pending, 0 when absent); cyc: complex [frames][76]; iq_last_frame: the stream's most recent frame from its first PRS
    sample.  Returns the new dict."""
    n = cyc.size
    frames = cyc.shape[0]
    new = dict(state)
    x = np.asarray(iq_last_frame[:4096])
    l1 = np.float32((np.abs(x.real).astype(np.float64) + np.abs(x.imag).astype(np.float64)).sum() / 4096.0)
    level_lost = bool(state["signal_average"] > 0 and l1 < np.float32(thr_null_start) * np.float32(state["signal_average"]))
    steers = not (level_lost and frames == 1)                    # one frame whose level is gone steers nothing
    if dd:                                                       # `cyc` holds dd4 sums: the decision-directed loop, gated
        err, gate_state = dd_loop_error(cyc, state, dd_gate, terms_per_frame)
        if steers:                                               # (an estimate that is not applied leaves the gate's memory alone)
            new.update(gate_state)
    else:
        err = np.float32(np.angle(cyc.astype(np.complex128)).sum() / n / (2.0 * np.pi * 2048.0))
    if steers:
        half = np.float32(0.5 / 2048.0)
        f = np.float32(state["fine_freq_offset"]) - np.float32(beta) * err
        if f > half:
            f -= 2 * half
        if f < -half:
            f += 2 * half
        new["fine_freq_offset"] = np.float32(f)
    new["last_fine_error"] = err
    if level_lost:
        new["total_frames_desync"] = state["total_frames_desync"] + 1
        new["total_frames_read"] = state["total_frames_read"] + frames - 1
    else:
        new["total_frames_read"] = state["total_frames_read"] + frames
        sb = np.float32(signal_beta)
        new["signal_average"] = (sb * np.float32(state["signal_average"]) + (np.float32(1.0) - sb) * l1) if state["signal_average"] > 0 else l1
    return new


# ---- the SIMD port (oracle/simd_port.c): cpu_baseline's second implementation, NOT the oracle ----------------------
_SIMD = None


def simd_lib():
    """libsimdport built `-O3 -march=native -ffast-math` FOR THIS HOST: the object is keyed by the CPU's feature flags
    (oracle/_native/libsimdport_<hash>.so), so one built on another machine is never loaded here."""
    global _SIMD
    if _SIMD is None:
        import hashlib
        try:
            flags = [l for l in open("/proc/cpuinfo") if l.startswith("flags")][0]
        except Exception:
            flags = "unknown"
        src = [os.path.join(_HERE, f) for f in ("simd_port.c", "oracle_bench.c", "dab_oracle.c", "dab_oracle.h")]
        key = hashlib.sha1((flags + "".join(str(os.path.getmtime(f)) for f in src)).encode()).hexdigest()[:12]
        out_dir = os.path.join(_HERE, "_native")
        so = os.path.join(out_dir, "libsimdport_%s.so" % key)
        if not os.path.exists(so):
            try:
                os.makedirs(out_dir, exist_ok=True)
                tmp = so + ".%d.tmp" % os.getpid()
                subprocess.check_call(["make", "-C", _HERE, "simd", "SIMD_OUT=" + tmp], stdout=subprocess.DEVNULL)
                os.replace(tmp, so)
            except Exception:                                   # read-only tree: build under the temporary directory
                import tempfile
                so = os.path.join(tempfile.gettempdir(), "libsimdport_%s_%d.so" % (key, os.getuid()))
                if not os.path.exists(so):
                    subprocess.check_call(["make", "-C", _HERE, "simd", "SIMD_OUT=" + so], stdout=subprocess.DEVNULL)
        _SIMD = C.CDLL(so)
        for n in ("bench_frames", "bench_frames_timed", "bench_ofdm_only_timed", "bench_pipeline_timed"):
            getattr(_SIMD, "simd_" + n).restype = C.c_double
        _SIMD.simd_port_isa.restype = C.c_char_p
    return _SIMD


def simd_isa():
    return simd_lib().simd_port_isa().decode()


def simd_ofdm_demod_frame(iq, freq_offset=0.0):
    a = np.ascontiguousarray(iq, np.complex64)
    assert a.size >= NB_SYMBOLS * NB_SYM
    soft = np.zeros(NB_FRAME_BITS, np.int8)
    simd_lib().simd_ofdm_demod_frame(_p(a), C.c_float(freq_offset), _p(soft))
    return soft


def simd_fic_decode(soft):
    s = np.ascontiguousarray(soft[:NB_FIC_BITS], np.int8)
    fib = np.zeros((12, 32), np.uint8)
    ok = np.zeros(12, np.uint8)
    simd_lib().simd_fic_decode(_p(s), _p(fib), _p(ok))
    return fib, ok


def simd_msc_decode_lf(deint, mask, nsteps):
    d = np.ascontiguousarray(deint, np.int8)
    m = np.ascontiguousarray(mask, np.uint8)
    out = np.zeros((nsteps - 6) // 8, np.uint8)
    simd_lib().simd_msc_decode_lf(_p(d), _p(m), C.c_int(nsteps), _p(out))
    return out


def _bench_fn(name, simd):
    return getattr(simd_lib(), "simd_" + name) if simd else getattr(lib(), "oracle_" + name)


def bench_frames(iq, freq_offset, total, threads, mask, nsteps, sc_bits):
    """Wall seconds for `total` frames of the whole per-frame hot path over `threads` pthreads.
    iq: complex64 [n_frames][76*2552]."""
    a = np.ascontiguousarray(iq, np.complex64)
    fo = np.ascontiguousarray(freq_offset, np.float32)
    m = np.ascontiguousarray(mask, np.uint8)
    return float(lib().oracle_bench_frames(_p(a), C.c_size_t(a.shape[1]), _p(fo), C.c_int(a.shape[0]), C.c_int(total),
                                           C.c_int(threads), _p(m), C.c_int(nsteps), C.c_int(sc_bits)))


def bench_frames_timed(iq, freq_offset, seconds, threads, mask, nsteps, sc_bits, simd=False):
    """Run the per-frame hot path on `threads` pthreads for about `seconds`; returns (frames_done, elapsed_s).
    simd=True: the SIMD port instead of the oracle (the same harness, oracle_bench.c compiled with -DBENCH_SIMD)."""
    a = np.ascontiguousarray(iq, np.complex64)
    fo = np.ascontiguousarray(freq_offset, np.float32)
    m = np.ascontiguousarray(mask, np.uint8)
    done = C.c_long(0)
    el = _bench_fn("bench_frames_timed", simd)(_p(a), C.c_size_t(a.shape[1]), _p(fo), C.c_int(a.shape[0]), C.c_double(seconds),
                                         C.c_int(threads), _p(m), C.c_int(nsteps), C.c_int(sc_bits), C.byref(done))
    return int(done.value), float(el)


def bench_ofdm_only_timed(iq, freq_offset, seconds, simd=False):
    """BASELINE config 1: the front end alone (A2..A6) on one thread for about `seconds`; (frames_done, elapsed_s)."""
    a = np.ascontiguousarray(iq, np.complex64)
    fo = np.ascontiguousarray(freq_offset, np.float32)
    done = C.c_long(0)
    el = _bench_fn("bench_ofdm_only_timed", simd)(_p(a), C.c_size_t(a.shape[1]), _p(fo), C.c_int(a.shape[0]), C.c_double(seconds),
                                            C.byref(done))
    return int(done.value), float(el)


def bench_pipeline_timed(iq, freq_offset, seconds, mask, nsteps, sc_bits, simd=False):
    """The plugin's deployment shape: one OFDM thread feeding one decoder thread through a two-frame ring
    (/root/reference/src/dab_module.cpp:92, src/radio_block.cpp:23-44); (frames_decoded, elapsed_s)."""
    a = np.ascontiguousarray(iq, np.complex64)
    fo = np.ascontiguousarray(freq_offset, np.float32)
    m = np.ascontiguousarray(mask, np.uint8)
    done = C.c_long(0)
    el = _bench_fn("bench_pipeline_timed", simd)(_p(a), C.c_size_t(a.shape[1]), _p(fo), C.c_int(a.shape[0]), C.c_double(seconds),
                                           _p(m), C.c_int(nsteps), C.c_int(sc_bits), C.byref(done))
    return int(done.value), float(el)


def sync_prs(sym, freq_offset=0.0, max_coarse=200, expected=0, distance_prob=1.0, first_path_rel=0.0):
    """sym: complex64[2552] from the candidate PRS start -> (coarse_carriers, time_offset, peak_to_mean, coarse_ptm).
    expected / distance_prob / first_path_rel: the tap choice of oracle_sync_prs_ex (defaults = strongest tap)."""
    a = np.ascontiguousarray(sym, np.complex64)
    assert a.size >= NB_SYM
    k, t = C.c_int32(0), C.c_int32(0)
    p, cp = C.c_float(0), C.c_float(0)
    lib().oracle_sync_prs_ex(_p(a), C.c_float(freq_offset), C.c_int(max_coarse), C.c_int(expected), C.c_float(distance_prob),
                             C.c_float(first_path_rel), C.byref(k), C.byref(t), C.byref(p), C.byref(cp))
    return k.value, t.value, p.value, cp.value


class AcquiredFrame(C.Structure):
    """== oracle_acquired_frame == dabgpu_acquired_frame (32 bytes)."""
    _fields_ = [("start", C.c_int64), ("freq_offset", C.c_float), ("coarse_carriers", C.c_int32),
                ("fine_offset", C.c_float), ("peak_to_mean", C.c_float), ("coarse_peak_to_mean", C.c_float),
                ("flags", C.c_int32)]


def null_block_l1(iq):
    a = np.ascontiguousarray(iq, np.complex64).ravel()
    out = np.zeros(a.size // 64, np.float32)
    lib().oracle_null_block_l1(_p(a), C.c_int64(a.size), _p(out))
    return out


def null_search(iq, thr_start=0.35, thr_end=0.75, min_blocks=30, max_out=64, level_chunk=256):
    """Candidate PRS starts (sample indices) of every whole frame in an unaligned capture.  level_chunk: blocks per
    chunk of the local level estimate (0 = thresholds relative to the capture's mean); default = dabgpu_acquire_default_cfg's."""
    a = np.ascontiguousarray(iq, np.complex64).ravel()
    out = np.zeros(max_out, np.int64)
    lib().oracle_null_search_ex.restype = C.c_int
    n = lib().oracle_null_search_ex(_p(a), C.c_int64(a.size), C.c_float(thr_start), C.c_float(thr_end), C.c_int(min_blocks),
                                    C.c_int(level_chunk), C.c_int(max_out), _p(out))
    return out[:n]


def acquire_candidate(iq, cand, max_coarse=200, min_peak_to_mean=30.0, margin=0, distance_prob=0.15, first_path_rel=0.25):
    """(defaults of the tap choice = dabgpu_acquire_default_cfg's)"""
    a = np.ascontiguousarray(iq, np.complex64).ravel()
    r = AcquiredFrame()
    lib().oracle_acquire_candidate_ex(_p(a), C.c_int64(a.size), C.c_int64(int(cand)), C.c_int(max_coarse),
                                      C.c_float(min_peak_to_mean), C.c_int(margin), C.c_float(distance_prob),
                                      C.c_float(first_path_rel), C.byref(r))
    return r


# ---- timing tracking (TEST INFRASTRUCTURE; parity unpinned) ------------------------------------------------------
# Restates what the reference's OFDM_Demod does in RUNNING_FINE_TIME_SYNC on every frame once locked
# (/root/reference/src/render_radio_block.cpp:196; knobs :224-225) in the batch form of dabgpu_ofdm_demod_tracked_dev:
# positions predicted from a per-stream state, each frame synchronised on its own PRS, the state moved on by a line
# through the measured starts.  The DSP lives in the absent DAB-Radio sub-module; this is the form the oracle fixes.
FRAME_LEN = 76 * 2552
L_FRAME = 196608


def track_predict(state, n_samples, max_frames):
    """-> (j0, [candidate start of slot i for the slots inside the capture]) from state['next_frame_start'], ['drift']"""
    nxt = float(state["next_frame_start"])
    period = float(L_FRAME) + float(np.float32(state["drift"]))
    j0 = int(np.ceil(-nxt / period)) if nxt < 0.0 else 0
    cands = []
    for i in range(max_frames):
        c = int(np.rint(nxt + float(j0 + i) * period))
        if c < 0 or c + FRAME_LEN + 512 > n_samples:
            break
        cands.append(c)
    return j0, cands


def track_sync(iq, state, max_frames, margin=64, min_peak_to_mean=100.0, distance_prob=0.15, first_path_rel=0.25):
    """Frames of one stream's capture `iq` (complex64 [n_samples]) as dabgpu_ofdm_demod_tracked_dev finds them:
    -> list of dicts (start, freq_offset, peak_to_mean, flags, time_offset), one per slot inside the capture."""
    a = np.ascontiguousarray(iq, np.complex64).ravel()
    out = []
    if not state["tracking"]:
        return out
    _, cands = track_predict(state, a.size, max_frames)
    fo = np.float32(state["fine_freq_offset"]) + np.float32(state["coarse_freq_offset"])
    for c in cands:
        _, toff, ptm, _ = sync_prs(a[c:c + NB_SYM], float(fo), 0, expected=margin, distance_prob=distance_prob,
                                   first_path_rel=first_path_rel)
        start = c + toff - margin
        flags = (1 if ptm >= min_peak_to_mean else 0) | (2 if (start >= 0 and start + FRAME_LEN <= a.size) else 0)
        out.append({"start": start, "freq_offset": fo, "peak_to_mean": ptm, "flags": flags, "time_offset": toff, "cand": c})
    return out


def track_update(state, frames, cyc, iq, n_samples, max_frames, advance, fine_beta=0.9, drift_beta=0.5, signal_beta=0.95,
                 thr_null_start=0.35, dd=False, dd_gate=2.5, terms_per_frame=19200):
    """State after a tracked call: `frames` from track_sync, `cyc` complex [len(frames)][76] of the demodulated frames
    (rows of unlocked frames are ignored), `iq` the capture.  Returns (new state dict, count)."""
    st = dict(state)
    if not state["tracking"]:
        return st, 0
    nxt = float(state["next_frame_start"])
    period = float(L_FRAME) + float(np.float32(state["drift"]))
    j0, cands = track_predict(state, n_samples, max_frames)
    count = len(cands)
    locked = [i for i in range(count) if (frames[i]["flags"] & 3) == 3]
    n = len(locked)
    desync = j0 + (count - n)
    level_lost = False
    if n > 0:
        last = locked[-1]
        x = np.asarray(iq[frames[last]["start"]:frames[last]["start"] + 4096])
        l1 = np.float32((np.abs(x.real).astype(np.float64) + np.abs(x.imag).astype(np.float64)).sum() / 4096.0)
        level_lost = bool(state["signal_average"] > 0 and l1 < np.float32(thr_null_start) * np.float32(state["signal_average"]))
        steers = not (level_lost and n == 1)                     # one locked frame whose level is gone steers nothing
        if dd:
            err, gate_state = dd_loop_error(np.asarray(cyc)[locked], state, dd_gate, terms_per_frame)
            if steers:                                           # (an estimate that is not applied leaves the gate's memory alone)
                st.update(gate_state)
        else:
            ang = np.angle(np.asarray(cyc)[locked].astype(np.complex128)).sum()
            err = np.float32(np.float32(ang / (n * 76.0)) * np.float32(1.0 / (6.283185307179586 * 2048.0)))
        if steers:
            half = np.float32(0.5 / 2048.0)
            f = np.float32(state["fine_freq_offset"]) - np.float32(fine_beta) * err
            if f > half:
                f -= 2 * half
            if f < -half:
                f += 2 * half
            st["fine_freq_offset"] = np.float32(f)
        st["last_fine_error"] = err
        st["last_time_offset"] = int(frames[last]["start"] - int(np.rint(nxt + float(j0 + last) * period)))
        st["last_peak_to_mean"] = np.float32(frames[last]["peak_to_mean"])
        if level_lost:
            desync += 1
        else:
            sb = np.float32(signal_beta)
            st["signal_average"] = (sb * np.float32(state["signal_average"]) + (np.float32(1) - sb) * l1) if state["signal_average"] > 0 else l1
    alpha = slope = slope_hat = gain = 0.0
    if n >= 2:
        i = np.array(locked, np.float64)
        r = np.array([float(frames[k]["start"]) - (nxt + float(j0 + k) * period) for k in locked], np.float64)
        sn, si, sii, sr, sir = float(n), i.sum(), (i * i).sum(), r.sum(), (i * r).sum()
        slope = (sn * sir - si * sr) / (sn * sii - si * si)
        alpha = (sr - slope * si) / sn
        slope_hat = slope
        gain = float(np.float32(drift_beta)) * min(1.0, n * 0.25)
    elif n == 1:
        k = locked[0]
        alpha = float(frames[k]["start"]) - (nxt + float(j0 + k) * period)
        slope_hat = alpha if count == 1 else 0.0
        gain = float(np.float32(drift_beta)) * 0.125
    if count > 0 and n == 0:
        st["tracking"] = 0
    st["next_frame_start"] = nxt + float(j0 + count) * period + alpha + slope * count - float(advance)
    st["drift"] = np.float32(float(np.float32(state["drift"])) + gain * slope_hat)
    st["total_frames_read"] = state["total_frames_read"] + n - (1 if level_lost else 0)   # (gone, not read)
    st["total_frames_desync"] = state["total_frames_desync"] + desync
    return st, count


def track_start(state, frames, advance):
    """State after dabgpu_track_start_dev: `frames` = the acquisition result of one stream (list of AcquiredFrame)."""
    st = dict(state)
    locked = [f for f in frames if (f.flags & 3) == 3]
    if not locked:
        st["tracking"] = 0
        return st
    s0 = locked[0].start
    y = np.array([float(f.start - s0) for f in locked], np.float64)
    j = np.rint(y / float(L_FRAME))
    n = float(len(locked))
    drift = 0.0
    if n >= 4:
        drift = (n * (j * y).sum() - j.sum() * y.sum()) / (n * (j * j).sum() - j.sum() ** 2) - float(L_FRAME)
    st["drift"] = np.float32(drift)
    st["next_frame_start"] = float(locked[-1].start) + float(L_FRAME) + float(np.float32(drift)) - float(advance)
    st["fine_freq_offset"] = np.float32(sum(float(f.fine_offset) for f in locked) / n)
    st["coarse_freq_offset"] = np.float32(-float(locked[-1].coarse_carriers) / 2048.0)
    st["tracking"] = 1
    st["dd_branch"] = st["dd_pending"] = DD_NO_BRANCH           # pulling in again: the first estimate is believed on any branch
    st["total_frames_read"] = state["total_frames_read"] + int(n)
    st["last_time_offset"] = 0
    st["last_peak_to_mean"] = np.float32(locked[-1].peak_to_mean)
    return st


def firecode(data):
    a = np.ascontiguousarray(data, np.uint8)
    return int(lib().oracle_firecode(_p(a), C.c_int(a.size)))


def rs_encode(data110):
    d = np.ascontiguousarray(data110, np.uint8)
    assert d.size == 110
    out = np.zeros(10, np.uint8)
    lib().oracle_rs_encode(_p(d), _p(out))
    return out


def rs_decode(cw120):
    """-> (corrected codeword, n_corrected or -1)."""
    c = np.array(cw120, np.uint8, copy=True)
    assert c.size == 120
    r = lib().oracle_rs_decode(_p(c))
    return c, int(r)


def dabplus_superframe(sf, s):
    """sf: uint8[120*s] -> (corrected sf, status[5], au_start[8])."""
    a = np.array(sf, np.uint8, copy=True)
    assert a.size == 120 * s
    st = np.zeros(5, np.int32)
    au = np.zeros(8, np.int32)
    lib().oracle_dabplus_superframe(_p(a), C.c_int(s), _p(st), _p(au))
    return a, st, au
