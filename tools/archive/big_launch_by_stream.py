#!/usr/bin/env python3
"""Does a big HBM-bound launch depend on the stream (hardware queue) it is enqueued on?  The front end (4096 frames) and
the mover of its geometry on 8 torch streams of one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, dabgpu
n = 4096
dev = torch.device("cuda", 0)
iq = torch.randn((n, 196608, 2), dtype=torch.float32, device=dev)
fo = ((torch.rand(n, device=dev) - 0.5) * 0.8 / 2048).float()
soft = torch.empty((n, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev)
cyc = torch.zeros((n, 76, 2), dtype=torch.float32, device=dev)
ctx = dabgpu.Context(0, n)
streams = [torch.cuda.Stream() for _ in range(8)]
torch.cuda.synchronize()
def t(fn, st, reps=6):
    with torch.cuda.stream(st):
        for _ in range(2): fn(st.cuda_stream)
        st.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps): fn(st.cuda_stream)
        e1.record(st); st.synchronize()
    return e0.elapsed_time(e1) / reps
for rnd in range(2):
    for i, st in enumerate(streams):
        a = t(lambda s: ctx.ofdm_demod_frames_dev(iq.data_ptr() + 2656 * 8, 196608, n, fo.data_ptr(), soft.data_ptr(), cyc.data_ptr(), None, s), st)
        b = t(lambda s: ctx.mover_frames_dev(iq.data_ptr() + 2656 * 8, 196608, n, soft.data_ptr(), True, s), st)
        print("round %d stream %d: front end %.3f ms  mover %.3f ms" % (rnd, i, a, b), flush=True)
