#!/usr/bin/env python3
"""Time the OFDM kernels alone (GPU box). usage: tools/ofdm_time.py [n_frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, dabgpu
n = int(sys.argv[1]) if len(sys.argv) > 1 else int(os.environ.get("OFDM_TIME_N", "1024"))
dev = torch.device("cuda", 0)
iq = torch.randn((n, 196608, 2), dtype=torch.float32, device=dev)
fo = ((torch.rand(n, device=dev) - 0.5) * 0.8 / 2048).float()
soft = torch.empty((n, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev)
cyc = torch.zeros((n, 76, 2), dtype=torch.float32, device=dev)
spec = torch.empty((n, 76, 2048, 2), dtype=torch.float32, device=dev)
ctx = dabgpu.Context(0, n); st = torch.cuda.Stream(); torch.cuda.set_stream(st); s = st.cuda_stream
d_iq = iq.data_ptr() + 2656 * 8
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
f = t(lambda: ctx.ofdm_demod_frames_dev(d_iq, 196608, n, fo.data_ptr(), soft.data_ptr(), cyc.data_ptr(), None, s))
f0 = t(lambda: ctx.ofdm_demod_frames_dev(d_iq, 196608, n, None, soft.data_ptr(), None, None, s))
g = t(lambda: ctx.fft_symbols_dev(d_iq, 196608, n, fo.data_ptr(), spec.data_ptr(), s))
print("lib %s fused us %.1f (%.0f GB/s)  fused(no pll,no cyc) %.1f  fft-only %.1f (%.0f GB/s)" % (
    os.environ.get("DABGPU_LIB", "default"), f, 1782016 * n / f / 1e3, f0, g, 2796800 * n / g / 1e3))
