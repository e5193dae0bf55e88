#!/usr/bin/env python3
"""Fused OFDM kernel: aligned-frame API vs acquired-frame API at even and odd sample offsets (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, dabgpu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
iq = torch.randn((n, 196608 + 8, 2), dtype=torch.float32, device=dev)
fo = ((torch.rand(n, device=dev) - 0.5) * 0.8 / 2048).float()
soft = torch.empty((n, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev)
cyc = torch.zeros((n, 76, 2), dtype=torch.float32, device=dev)
ctx = dabgpu.Context(0, n); st = torch.cuda.Stream(); torch.cuda.set_stream(st); s = st.cuda_stream
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
stride = 196608 + 8
for with_cyc in (True, False):
    c = cyc.data_ptr() if with_cyc else None
    a = t(lambda: ctx.ofdm_demod_frames_dev(iq.data_ptr() + 2656 * 8, stride, n, fo.data_ptr(), soft.data_ptr(), c, None, s))
    res = [a]
    for off in (2656, 2657):
        fr = np.zeros(n, dabgpu.ACQUIRED_FRAME_DTYPE)
        fr["start"] = off; fr["flags"] = 3; fr["freq_offset"] = fo.cpu().numpy()
        d_fr = torch.from_numpy(fr.view(np.uint8).reshape(n, 32)).to(dev)
        res.append(t(lambda: ctx.ofdm_demod_acquired_dev(iq.data_ptr(), stride, n, 1, d_fr.data_ptr(), soft.data_ptr(), c, None, s)))
    print("cyc=%d  frames API %.1f us   acquired even %.1f us   acquired odd %.1f us   (%d frames)" % (with_cyc, *res, n))
