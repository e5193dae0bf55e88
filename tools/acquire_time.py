#!/usr/bin/env python3
"""Time acquisition + in-place demodulation of unaligned captures (GPU box).
usage: tools/acquire_time.py [n_streams] [frames_per_stream]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, dabgpu
from dabgpu import synth
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
F = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda", 0)
e = synth.Ensemble(seed=1, n_frames=4)
base = torch.from_numpy(e.iq().ravel()).to(dev)                       # 4 frames
n = F * 196608 + 100000
iq = torch.empty((S, n), dtype=torch.complex64, device=dev)
g = torch.Generator(device=dev); g.manual_seed(3)
t = torch.arange(n, device=dev, dtype=torch.float64)
for s in range(S):
    cut = 1001 + 7919 * s % 190000
    reps = (n + cut) // base.numel() + 2
    x = base.repeat(reps)[cut:cut + n]
    cfo = ((s % 7) - 3 + 0.1 * (s % 5)) / 2048.0
    noise = torch.randn(n, generator=g, device=dev) + 1j * torch.randn(n, generator=g, device=dev)
    iq[s] = x * torch.exp(2j * np.pi * cfo * t).to(torch.complex64) + 0.1 * noise
M = F + 1
frames = torch.zeros((S, M, 32), dtype=torch.uint8, device=dev)
counts = torch.zeros(S, dtype=torch.int32, device=dev)
soft = torch.zeros((S * M, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev)
fib = torch.zeros((S * M, 12, 32), dtype=torch.uint8, device=dev); ok = torch.zeros((S * M, 12), dtype=torch.uint8, device=dev)
ctx = dabgpu.Context(0, 64); st = torch.cuda.Stream(); torch.cuda.set_stream(st); s_ = st.cuda_stream
def t_(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
ta = t_(lambda: ctx.acquire_dev(iq.data_ptr(), n, S, n, M, frames.data_ptr(), counts.data_ptr(), None, s_))
td = t_(lambda: ctx.ofdm_demod_acquired_dev(iq.data_ptr(), n, S, M, frames.data_ptr(), soft.data_ptr(), None, None, s_))
ctx.fic_decode_dev(soft.data_ptr(), dabgpu.NB_FRAME_BITS, S * M, fib.data_ptr(), ok.data_ptr(), s_); ctx.sync()
cnt = counts.cpu().numpy(); okh = ok.cpu().numpy().reshape(S, M, 12)
fr = frames.cpu().numpy().view(dabgpu.ACQUIRED_FRAME_DTYPE).reshape(S, M)
nf = int(cnt.sum()); good = sum(int(okh[s, :cnt[s]].all(axis=1).sum()) for s in range(S))
odd = sum(int((fr[s, :cnt[s]]["start"] & 1).sum()) for s in range(S))
print("streams %d x %.2f s capture (%.1f GB): found %d frames (%d at odd sample offsets), %d with all 12 FIB CRCs ok" % (S, n / 2.048e6, S * n * 8 / 1e9, nf, odd, good))
print("acquire %.3f ms (%.0f GB/s of capture)   demod-in-place %.3f ms = %.0f frames/s (%.0f GB/s algorithmic)" % (
    ta, S * n * 8 / ta / 1e6, td, nf / td * 1e3, nf * 1782016 / td / 1e6))
