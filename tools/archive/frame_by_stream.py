#!/usr/bin/env python3
"""The plugin's two one-frame calls on each of eight contexts (= eight streams, spread by the runtime over its hardware queues) of one
process: does a call's time depend on WHICH stream runs it?  (It did: profiles/r05_frame_path.md, "the stream lottery".)
Under `rocprofv3 --kernel-trace`, tools/frame_by_queue_report.py then gives every kernel's mean duration per hardware queue."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np, dabgpu
from dabgpu import synth
rng = np.random.default_rng(0)
e = synth.Ensemble(seed=1, n_frames=4)
n_host, M, used = 32, 64, 76 * 2552
rx = synth.channel(np.tile(e.iq().ravel(), (n_host + 4) // 4 + 1), snr_db=20.0, cfo=0.2 / 2048, rng=rng)
L = synth.NB_FRAME_SAMPLES
lib = dabgpu.lib()
p_iq = dabgpu.PinnedArray((n_host, used), np.complex64)
for f in range(n_host):
    lo = f * L + synth.NB_NULL - M
    p_iq.array[f] = rx[lo:lo + used]
p_soft = dabgpu.PinnedArray((dabgpu.NB_FRAME_BITS,), np.int8)
p_fib = dabgpu.PinnedArray((1, 12, 32), np.uint8); p_ok = dabgpu.PinnedArray((1, 12), np.uint8); p_out = dabgpu.PinnedArray((4, 192), np.uint8)
sc = dabgpu.subchannel(0, 64, level=3); arr = (dabgpu.Subchannel * 1)(sc); outs = (C.c_void_p * 1)(p_out.array.ctypes.data)
cfg = dabgpu.track_cfg(timing_margin=M); fres = dabgpu.FrameResult()

def run(co, label):
    co.streams_reset(1)
    t, td = [], []
    for rep in range(3):
        for f in range(n_host):
            t0 = time.perf_counter()
            rc = lib.dabgpu_ofdm_demod_stream_frame(co._h, 0, p_iq.array[f].ctypes.data, 1 if (rep == 0 and f == 0) else 0, C.byref(cfg),
                                                    p_soft.array.ctypes.data, None, C.byref(fres))
            t1 = time.perf_counter()
            rc2 = lib.dabgpu_decode_stream_frames(co._h, p_soft.array.ctypes.data, dabgpu.NB_FRAME_BITS, 1, p_fib.array.ctypes.data,
                                                  p_ok.array.ctypes.data, arr, 1, outs)
            t2 = time.perf_counter()
            assert rc == 0 and rc2 == 0 and fres.flags == 3
            if rep > 0: t.append(t1 - t0); td.append(t2 - t1)
    print("%-40s frame call %.1f us   decode call %.1f us" % (label, np.mean(t) * 1e6, np.mean(td) * 1e6), flush=True)

ctxs = [dabgpu.Context(device=0, max_frames=1) for _ in range(8)]
for i, c in enumerate(ctxs):
    run(c, "context %d of 8 (creation order)" % i)
for i in (0, 1, 2):
    run(ctxs[i], "again: context %d" % i)
for c in ctxs[1:]:
    c.close()
run(ctxs[0], "context 0 after the others are closed")
c9 = dabgpu.Context(device=0, max_frames=1)
run(c9, "a new context after that")
