#!/usr/bin/env python3
"""Does the grouped decoder care where its survivor scratch sits relative to the soft bits?  The scratch is allocated at
the first decode call of a context; a dummy allocation of varying size before that call moves it through the address
space.  usage: tools/scratch_place.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import torch, dabgpu
dev = torch.device("cuda", 0)
E, F = 64, 256; n = E * F; NB = dabgpu.NB_FRAME_BITS
st = torch.cuda.Stream(); torch.cuda.set_stream(st); s = st.cuda_stream
soft = torch.randint(-127, 128, (n, NB), dtype=torch.int8, device=dev)
fib = torch.zeros((n, 12, 32), dtype=torch.uint8, device=dev); crc = torch.zeros((n, 12), dtype=torch.uint8, device=dev)
sc = dabgpu.subchannel(0, 64, level=3)
msc = torch.zeros((E, F * 4, 192), dtype=torch.uint8, device=dev)
for gb in (0, 16, 32, 48, 64, 80, 96, 128, 160):
    dummy = torch.empty((gb << 30,), dtype=torch.uint8, device=dev) if gb else None
    c = dabgpu.Context(0, n)
    def run(): c.decode_frames_dev(soft.data_ptr(), NB, E, F, fib.data_ptr(), crc.data_ptr(), [sc], None, None, [msc.data_ptr()], s)
    for _ in range(3): run()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    print("dummy %3d GB before the scratch: %.3f ms per decode" % (gb, e0.elapsed_time(e1) / 10))
    c.close(); del dummy; torch.cuda.empty_cache()
