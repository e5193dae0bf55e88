#!/usr/bin/env python3
"""One ensemble on the GPU (BASELINE configs 2 and 3): device time of each call of the step, per frames-per-call.
usage: tools/single_time.py [frames ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, dabgpu
from dabgpu import synth
dev = torch.device("cuda", 0); torch.manual_seed(7)
sizes = [int(a) for a in sys.argv[1:]] or [4, 16, 64, 256, 1024]
Fmax = max(sizes)
L = synth.NB_FRAME_SAMPLES if hasattr(synth, "NB_FRAME_SAMPLES") else 196608
iq = (torch.randn((Fmax * L, 2), device=dev) * 0.7).contiguous()
soft = torch.zeros((Fmax, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev)
fib = torch.zeros((Fmax, 12, 32), dtype=torch.uint8, device=dev); ok = torch.zeros((Fmax, 12), dtype=torch.uint8, device=dev)
sc = dabgpu.subchannel(0, 64, level=3)
msc = torch.zeros((1, Fmax * 4, 192), dtype=torch.uint8, device=dev)
# (a de-interleaver history of noise, as the soft bits: an all-erasure history is the traceback's worst case, not its usual one)
hin = torch.randint(-127, 128, (1, 15, sc.length * 64), dtype=torch.int8, device=dev); hout = torch.zeros_like(hin)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); s = st.cuda_stream
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("(front end on the default loop: all 76 prefixes read; files before r05_single_time.txt used the decision-directed one)")
print("frames  front_end_us  fic_us  fic+msc64_us  step(cfg2)_us  step(cfg3)_us   frames/s cfg2   cfg3")
for F in sizes:
    c = dabgpu.Context(0, F); c.streams_reset(1)          # (the library's default loop: cyclic-prefix correlations)
    fe = lambda: c.ofdm_demod_streams_dev(iq.data_ptr(), L, 1, F, 0.1, soft.data_ptr(), None, None, s)
    d2 = lambda: c.decode_frames_dev(soft.data_ptr(), dabgpu.NB_FRAME_BITS, 1, F, fib.data_ptr(), ok.data_ptr(), [], [], [], [], s)
    d3 = lambda: c.decode_frames_dev(soft.data_ptr(), dabgpu.NB_FRAME_BITS, 1, F, fib.data_ptr(), ok.data_ptr(), [sc],
                                     [hin.data_ptr()], [hout.data_ptr()], [msc.data_ptr()], s)
    a, b, d = timed(fe), timed(d2), timed(d3)
    s2 = timed(lambda: (fe(), d2())); s3 = timed(lambda: (fe(), d3()))
    print("%6d  %10.1f  %7.1f  %10.1f  %12.1f  %12.1f   %10.0f  %10.0f" % (F, a, b, d, s2, s3, F / s2 * 1e6, F / s3 * 1e6))
    c.close()
