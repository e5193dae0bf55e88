#!/usr/bin/env python3
"""The plugin's use: ONE frame at a time from host memory (BasicRadio::Process, /root/reference/src/radio_block.cpp:42)
for a full multiplex -- FIC + 18 sub-channels (816 CUs).  Before: one dabgpu_fic_decode and one dabgpu_msc_decode per
sub-channel, each uploading the same 230400 soft bits and synchronising (what the host mirror did in round 1).
After: one dabgpu_decode_frames (one upload, one synchronisation), page-locked buffers.  GPU box; prints both."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np, dabgpu
rng = np.random.default_rng(0)
scs, cu = [], 0
for br, lvl, k in ((64, 3, 10), (48, 3, 4), (32, 2, 3)):
    for _ in range(k):
        sc = dabgpu.subchannel(cu, br, level=lvl); scs.append(sc); cu += sc.length
scs.append(dabgpu.uep_subchannel(35, cu)); cu += scs[-1].length
soft = rng.integers(-127, 128, (1, dabgpu.NB_FRAME_BITS), dtype=np.int8)
ctx = dabgpu.Context(0, 1)
hist = [np.zeros((1, 15, sc.length * 64), np.int8) for sc in scs]
reps = 50


def before():
    ctx.fic_decode(soft)
    for k, sc in enumerate(scs):
        ctx.msc_decode(sc, soft, n_streams=1, history_in=hist[k], want_history=True)


def timeit(fn):
    for _ in range(5): fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e3


t_before = timeit(before)
# after: the C ABI directly on page-locked buffers, as the host mirror calls it
L = dabgpu.lib()
n = len(scs)
p_soft = dabgpu.PinnedArray(soft.shape, np.int8); p_soft.array[:] = soft
p_fib = dabgpu.PinnedArray((1, 12, 32), np.uint8); p_ok = dabgpu.PinnedArray((1, 12), np.uint8)
p_hi = [dabgpu.PinnedArray(h.shape, np.int8) for h in hist]
p_ho = [dabgpu.PinnedArray(h.shape, np.int8) for h in hist]
p_out = [dabgpu.PinnedArray((1, 4, sc.bitrate_kbps * 3), np.uint8) for sc in scs]
arr = (dabgpu.Subchannel * n)(*scs)
hi = (C.c_void_p * n)(*[a.array.ctypes.data for a in p_hi])
ho = (C.c_void_p * n)(*[a.array.ctypes.data for a in p_ho])
out = (C.c_void_p * n)(*[a.array.ctypes.data for a in p_out])


def after():
    assert L.dabgpu_decode_frames(ctx._h, p_soft.array.ctypes.data, dabgpu.NB_FRAME_BITS, 1, 1, p_fib.array.ctypes.data,
                                  p_ok.array.ctypes.data, arr, n, hi, ho, out) == 0


t_after = timeit(after)


def stream():
    assert L.dabgpu_decode_stream_frames(ctx._h, p_soft.array.ctypes.data, dabgpu.NB_FRAME_BITS, 1, p_fib.array.ctypes.data,
                                         p_ok.array.ctypes.data, arr, n, out) == 0


t_stream = timeit(stream)
fib, ok = ctx.fic_decode(soft)
assert (p_fib.array == fib).all()
print("one frame, FIC + %d sub-channels (%d CUs), host buffers: before %.2f ms (%d uploads of the frame, %d synchronisations)  "
      "dabgpu_decode_frames %.2f ms (1 upload, 1 synchronisation, histories over PCIe)  dabgpu_decode_stream_frames %.3f ms "
      "(histories on the device, 1 upload + 1 download) = %.0f x real time" % (n, cu, t_before, n + 1, n + 1, t_after, t_stream,
                                                                                96.0 / t_stream))

# ---- the front end of the same use: one frame of IQ from host memory through OFDM_Demod's per-frame work ----
# round 2: three synchronous calls (dabgpu_sync_prs, dabgpu_ofdm_demod_streams, dabgpu_get_stats: the PRS crosses PCIe
# twice, three stream synchronisations); round 3: dabgpu_ofdm_demod_stream_frame (one upload, one download, one
# synchronisation, everything in between on the device)
from dabgpu import synth
e = synth.Ensemble(seed=1, n_frames=4)
iq = synth.channel(e.iq().ravel(), snr_db=20.0, cfo=0.2 / 2048, rng=rng)
M = 128
frame = np.ascontiguousarray(iq[synth.NB_NULL - M:synth.NB_NULL - M + 76 * 2552])
p_iq = dabgpu.PinnedArray(frame.shape, np.complex64); p_iq.array[:] = frame
p_sft = dabgpu.PinnedArray((dabgpu.NB_FRAME_BITS,), np.int8)
p_dq = dabgpu.PinnedArray((75, 1536), np.complex64)
octx = dabgpu.Context(0, 1); octx.streams_reset(1)
res3 = dabgpu.SyncResult(); st3 = dabgpu.Stats()
fine = C.c_float(0.0)


def three_calls(dq):
    assert L.dabgpu_sync_prs(octx._h, p_iq.array.ctypes.data, 76 * 2552, 1, C.byref(fine), 204, C.byref(res3)) == 0
    assert L.dabgpu_ofdm_demod_streams(octx._h, p_iq.array.ctypes.data, 76 * 2552, 1, 1, C.c_float(0.9), p_sft.array.ctypes.data,
                                       None, dq) == 0
    assert L.dabgpu_get_stats(octx._h, 0, C.byref(st3)) == 0


cfg = dabgpu.track_cfg(timing_margin=M)
fres = dabgpu.FrameResult()


def one_call(dq):
    assert L.dabgpu_ofdm_demod_stream_frame(octx._h, 0, p_iq.array.ctypes.data, 0, C.byref(cfg), p_sft.array.ctypes.data, dq,
                                            C.byref(fres)) == 0


for name, dq in (("soft bits only", None), ("+ the 75 x 1536 constellation the GUI plots (GetFrameDataVec)", p_dq.array.ctypes.data)):
    t3 = timeit(lambda: three_calls(dq))
    t1 = timeit(lambda: one_call(dq))
    print("one frame of IQ (1.55 MB) from page-locked host memory, %s: sync_prs + demod_streams + get_stats %.3f ms  ->  "
          "dabgpu_ofdm_demod_stream_frame %.3f ms = %.0f x real time" % (name, t3, t1, 96.0 / t1))
assert fres.flags == 3
