#!/usr/bin/env python3
"""Time the channel-decoder kernels alone (GPU box). usage: tools/vit_time.py [n_frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, dabgpu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0)
soft = torch.randint(-127, 128, (n, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev)
fib = torch.zeros((n, 12, 32), dtype=torch.uint8, device=dev); crc = torch.zeros((n, 12), dtype=torch.uint8, device=dev)
sc = dabgpu.subchannel(0, 64, level=3); E = 64; F = n // E
msc = torch.zeros((E, F * 4, 192), dtype=torch.uint8, device=dev)
ctx = dabgpu.Context(0, n); st = torch.cuda.Stream(); torch.cuda.set_stream(st); s = st.cuda_stream
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
print("lib", os.environ.get("DABGPU_LIB", "default"), "fic us %.1f" % t(lambda: ctx.fic_decode_dev(soft.data_ptr(), dabgpu.NB_FRAME_BITS, n, fib.data_ptr(), crc.data_ptr(), s)),
      "msc us %.1f" % t(lambda: ctx.msc_decode_dev(sc, soft.data_ptr(), dabgpu.NB_FRAME_BITS, E, F, None, None, msc.data_ptr(), s)))
