#!/usr/bin/env python3
"""GPU box: why does the channel decoder take 1.31 ms inside the bench's step and 1.06 ms on its own?  The decoder's launch pair
(64 x 256 frames, FIC + one 64 kbit/s sub-channel) timed (a) back to back with itself, (b) behind the fused front end, (c) behind
the front end's data mover (the same HBM traffic, no arithmetic), (d) behind the front end after a pause.
usage: python3 tools/clock_shadow.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, dabgpu
from dabgpu import synth
dev = torch.device("cuda", 0)
E, F = 64, 256
n, L, NB = E * F, synth.NB_FRAME_SAMPLES, dabgpu.NB_FRAME_BITS
ctx = dabgpu.Context(device=0, max_frames=n)
ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts); s = ts.cuda_stream
d_iq, d_soft, rep = ctx.alloc_frame_buffers(n, L, dabgpu.PLACE_DOMAINS)
iq = dabgpu.device_tensor(torch, d_iq, (n, L), torch.complex64, dev)
iq.view(torch.float32).normal_()
fo = torch.zeros((n,), dtype=torch.float32, device=dev)
cyc = torch.zeros((n, 76, 2), dtype=torch.float32, device=dev)
soft_in = torch.randint(-127, 128, (n, NB), dtype=torch.int8, device=dev)          # what the decoder reads (never overwritten)
fib = torch.zeros((n, 12, 32), dtype=torch.uint8, device=dev); crc = torch.zeros((n, 12), dtype=torch.uint8, device=dev)
sc = dabgpu.subchannel(0, 64, level=3)
msc = torch.zeros((E, F * 4, 192), dtype=torch.uint8, device=dev)
p_iq = d_iq + synth.NB_NULL * 8
def fe(): ctx.ofdm_demod_frames_dev(p_iq, L, n, fo.data_ptr(), d_soft, cyc.data_ptr(), None, s)
def mv(): ctx.mover_frames_dev(p_iq, L, n, d_soft, True, s)
def fe_nonco(): ctx.ofdm_demod_frames_dev(p_iq, L, n, None, d_soft, cyc.data_ptr(), None, s)       # -14 % instructions, the same bytes
def fe_nocyc(): ctx.ofdm_demod_frames_dev(p_iq, L, n, fo.data_ptr(), d_soft, None, None, s)        # no prefixes read: -17 % bytes
def dec(): ctx.decode_frames_dev(soft_in.data_ptr(), NB, E, F, fib.data_ptr(), crc.data_ptr(), [sc], [None], [None], [msc.data_ptr()], s)
def run(before, pause_s=0.0, steps=12):
    out, pre = [], []
    for k in range(steps):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        if before: before()
        e[1].record(); dec(); e[2].record()
        torch.cuda.synchronize()
        if k >= 2: out.append(e[1].elapsed_time(e[2])); pre.append(e[0].elapsed_time(e[1]))
        if pause_s: time.sleep(pause_s)
    return float(np.mean(pre)), float(np.mean(out))
for name, b, p in (("decoder back to back", None, 0.0), ("behind the fused front end", fe, 0.0), ("behind the front end's mover (no arithmetic)", mv, 0.0),
                   ("behind the front end without its frequency correction (-14 % instructions)", fe_nonco, 0.0),
                   ("behind the front end without the prefix correlations (-17 % bytes)", fe_nocyc, 0.0),
                   ("behind the fused front end (again)", fe, 0.0),
                   ("behind the fused front end, 50 ms pause between steps", fe, 0.05), ("decoder alone, 50 ms pause between launches", None, 0.05),
                   ("decoder back to back (again)", None, 0.0)):
    a, d = run(b, p)
    print("%-80s  before %.3f ms   decoder %.3f ms" % (name, a, d))
