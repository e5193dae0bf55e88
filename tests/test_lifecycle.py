"""Context lifecycle on the GPU: every scratch buffer, stage slot, ring and mapped chunk a context takes while it
works goes back when it is destroyed, and two contexts side by side do not see each other's stream states."""
import os

import numpy as np
import pytest

import dabgpu
from dabgpu import synth
from conftest import make_ctx

pytestmark = pytest.mark.gpu


def _exercise(c, frames, ensemble):
    """one pass over the entry points that allocate on demand"""
    import ctypes as C
    import torch
    n = frames.shape[0]
    L = frames.shape[1]
    dev = torch.device("cuda", 0)
    soft, cyc, _ = c.ofdm_demod_frames(frames, np.zeros(n, np.float32), want_cyc=True)
    fib, ok = c.fic_decode(soft)
    sc = dabgpu.subchannel(ensemble.start_cu, 64, level=3)
    c.msc_decode(sc, soft, n_streams=1, want_history=True)
    c.decode_frames(soft, 1, [sc])
    # streams: fixed stride, tracked, one frame from the host
    c.streams_reset(2)
    d_iq = torch.from_numpy(np.ascontiguousarray(frames[:4])).to(dev)
    d_soft = torch.zeros((4, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev)
    c.ofdm_demod_streams_dev(d_iq.data_ptr(), L, 2, 2, 0.9, d_soft.data_ptr(), None, None)
    c.set_stream_loop(decision_directed=True)
    c.ofdm_demod_streams_dev(d_iq.data_ptr(), L, 2, 2, 0.9, d_soft.data_ptr(), None, None)
    c.set_stream_loop(decision_directed=False)
    fr = torch.zeros((2, 4, 32), dtype=torch.uint8, device=dev)
    cnt = torch.zeros(2, dtype=torch.int32, device=dev)
    d_soft8 = torch.zeros((8, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev)
    c.ofdm_demod_tracked_dev(d_iq.data_ptr(), 2 * L, 2, 2 * L, 4, L, d_soft8.data_ptr(), fr.data_ptr(), cnt.data_ptr(),
                             cfg=dabgpu.track_cfg(auto_acquire=1))
    c.sync()
    res = dabgpu.FrameResult()
    cfg = dabgpu.track_cfg()
    host_soft = np.zeros(dabgpu.NB_FRAME_BITS, np.int8)
    one = np.ascontiguousarray(frames[0])
    assert dabgpu.lib().dabgpu_ofdm_demod_stream_frame(c._h, 0, one.ctypes.data, 1, C.byref(cfg), host_soft.ctypes.data, None,
                                                       C.byref(res)) == 0
    c.fft_symbols(frames[:1], np.zeros(1, np.float32))
    return ok


def test_contexts_give_everything_back(built, ensemble, ensemble_iq):
    import torch
    frames = np.ascontiguousarray(ensemble_iq[:, synth.NB_NULL:synth.NB_NULL + 76 * 2552])
    torch.cuda.synchronize()
    c = make_ctx(None, 8)                                    # first one: the runtime's own pools warm up
    assert _exercise(c, frames, ensemble).all()
    c.close()
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(12):
        c = make_ctx(None, 8)
        _exercise(c, frames, ensemble)
        c.close()
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    lost = free0 - torch.cuda.mem_get_info()[0]
    assert lost <= 32 << 20, lost                            # 12 contexts' worth of scratch would be several hundred MB


def test_two_contexts_keep_their_own_stream_states(built, ensemble, ensemble_iq):
    import torch
    dev = torch.device("cuda", 0)
    L = 76 * 2552
    a, b = make_ctx(None, 8), make_ctx(None, 8)
    a.streams_reset(1); b.streams_reset(1)
    rx = synth.channel(ensemble_iq.ravel(), snr_db=20.0, cfo=0.2 / 2048, rng=np.random.default_rng(5)).reshape(ensemble_iq.shape)
    frames = np.ascontiguousarray(rx[:2, synth.NB_NULL:synth.NB_NULL + L])
    d_iq = torch.from_numpy(frames).to(dev)
    cyc = torch.zeros((2, 76), dtype=torch.complex64, device=dev)
    soft = torch.zeros((2, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev)
    torch.cuda.synchronize()
    for _ in range(3):
        a.ofdm_demod_streams_dev(d_iq.data_ptr(), L, 1, 2, 0.9, soft.data_ptr(), cyc.data_ptr(), None)
    a.sync()
    sa, sb = a.get_stats(0), b.get_stats(0)
    assert abs(sa.fine_freq_offset * 2048 + 0.2) < 0.01 and sa.total_frames_read == 6
    assert sb.fine_freq_offset == 0 and sb.total_frames_read == 0
    a.close(); b.close()


def test_placed_allocations_repeat(built):
    """dabgpu_alloc_frame_buffers(PLACE_DOMAINS) / free, twenty times over IN THIS PROCESS: every call succeeds through
    the context's address ranges, reserved once (same addresses every round), the reports stay sane, and the device's free memory
    ends where it began.  (Round 3 ran this in a process of its own and excused a crash of the runtime's virtual-memory
    API; the allocator no longer frees and re-reserves address ranges, which is what provoked it.  tools/alloc_stress.py
    is the long version: hundreds of rounds beside a process that holds 100 GB.)"""
    import torch
    torch.cuda.synchronize()
    c = make_ctx(None, 8)
    L = dabgpu.NB_FRAME_SAMPLES
    d_iq, d_soft, rep = c.alloc_frame_buffers(3000, L, dabgpu.PLACE_DOMAINS)      # once, so that pools are warm
    # (a box whose virtual-memory chunks share one HBM domain ends in a plain pair every round: still through the ranges)
    assert rep.method == 1 or rep.fallback_reason == dabgpu.PLAIN_ONE_DOMAIN, rep.fallback_reason
    arena = d_iq if rep.method == 1 else None                  # every placed pair of a context lies at the same address
    c.free_frame_buffers(d_iq, d_soft)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    final = 3000 * (L * 8 + dabgpu.NB_FRAME_BITS)
    for k in range(20):
        d_iq, d_soft, rep = c.alloc_frame_buffers(3000, L, dabgpu.PLACE_DOMAINS)
        assert rep.method == 1 or rep.fallback_reason == dabgpu.PLAIN_ONE_DOMAIN, (k, rep.method, rep.fallback_reason)
        if rep.method == 1:
            assert arena in (None, d_iq), (k, arena, d_iq)
            arena = d_iq
        assert d_soft and 0 <= rep.conflicts <= 1000 and rep.runtime_error == 0
        assert rep.setup_peak_bytes <= 1.5 * final
        c.free_frame_buffers(d_iq, d_soft)
    torch.cuda.synchronize()
    assert abs(free0 - torch.cuda.mem_get_info()[0]) <= 64 << 20
    c.close()
