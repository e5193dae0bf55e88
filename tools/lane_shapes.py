#!/usr/bin/env python3
"""the lane decoder's tail imbalance.  The bench's launch puts ONE 774-step wave (64 FIC codewords) and
ONE 1542-step wave (64 sub-channel codewords) on every SIMD; the longer one runs alone for the last half of its life.
This times the grouped forward + traceback pair on launch shapes that isolate the rates involved (16 384 frames, noise):
    fic                 1 wave per SIMD, 774 steps              -> a lone wave's time per wave-step
    msc                 1 wave per SIMD, 1542 steps
    msc + msc           2 waves per SIMD, equal length           -> the two-wave rate with no tail
    fic + msc           the bench's shape
    fic + msc + msc     3 waves per SIMD: what "FIC wave + two half-occupancy sub-channel waves" would issue (a wave's
                        instruction count does not depend on how many of its lanes hold a codeword)
usage: tools/lane_shapes.py [n_frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, dabgpu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device("cuda", 0)
soft = torch.randint(-127, 128, (n, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev)
fib = torch.zeros((n, 12, 32), dtype=torch.uint8, device=dev); crc = torch.zeros((n, 12), dtype=torch.uint8, device=dev)
E = 64; F = n // E
a, b = dabgpu.subchannel(0, 64, level=3), dabgpu.subchannel(48, 64, level=3)
oa = torch.zeros((E, F * 4, 192), dtype=torch.uint8, device=dev); ob = torch.zeros_like(oa)
ctx = dabgpu.Context(0, n, flags=dabgpu.FLAG_VITERBI_LANE); st = torch.cuda.Stream(); torch.cuda.set_stream(st); s = st.cuda_stream
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
P = soft.data_ptr(); NB = dabgpu.NB_FRAME_BITS
rows = [
    ("fic", lambda: ctx.fic_decode_dev(P, NB, n, fib.data_ptr(), crc.data_ptr(), s)),
    ("msc", lambda: ctx.msc_decode_dev(a, P, NB, E, F, None, None, oa.data_ptr(), s)),
    ("msc + msc", lambda: ctx.msc_decode_multi_dev([a, b], P, NB, E, F, None, None, [oa.data_ptr(), ob.data_ptr()], s)),
    ("fic + msc", lambda: ctx.decode_frames_dev(P, NB, E, F, fib.data_ptr(), crc.data_ptr(), [a], None, None, [oa.data_ptr()], s)),
    ("fic + msc + msc", lambda: ctx.decode_frames_dev(P, NB, E, F, fib.data_ptr(), crc.data_ptr(), [a, b], None, None, [oa.data_ptr(), ob.data_ptr()], s)),
]
res = {}
for name, fn in rows:
    ctx.set_timing(True)
    us = t(fn)
    res[name] = us
    print("%-18s %8.1f us per call" % (name, us))
ws = n * 4 / 64 / 1024.0                                     # waves of one entry per SIMD (1.0 at 16 384 frames)
lone_f, lone_m, pair = res["fic"] / (774 * ws), res["msc"] / (1542 * ws), res["msc + msc"] / (2 * 1542 * ws)
print("per wave-step (call time / steps, traceback and launch included): lone FIC wave %.3f us, lone sub-channel wave %.3f us, "
      "two equal waves %.3f us each" % (lone_f, lone_m, pair))
print("bench shape: measured %.1f us; all %d wave-steps of a SIMD at the two-wave rate would take %.1f us; "
      "774 paired + 768 alone at the measured rates: %.1f us" % (res["fic + msc"], 774 + 1542, (774 + 1542) * pair * ws,
                                                                  (2 * 774 * pair + 768 * lone_m) * ws))
print("three waves per SIMD (FIC + two sub-channel waves): %.1f us = %.3f us per wave-step" % (res["fic + msc + msc"], res["fic + msc + msc"] / ((774 + 2 * 1542) * ws)))
