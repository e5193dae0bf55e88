#!/usr/bin/env python3
"""GPU box: the front end's data mover (dabgpu_mover_frames_dev, 1 024 frames: 1.6 GB read, 0.24 GB written) with its input window and
its output window at every pair of 8 GB steps of ONE plain allocation: where on this machine do a read stream and a write stream
get along?  On most boxes the cells fall into two values (~0.31 / ~0.35 ms: three 96 GB domains, profiles/r02_hbm_domains.txt); a
machine that "behaves as one domain" (profiles/r05_box_spread.txt) is the one this is for.
usage: python3 tools/domain_matrix.py [arena_GB=192] [step_GB=8]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, dabgpu
from dabgpu import synth
G = 1 << 30
arena_gb = int(sys.argv[1]) if len(sys.argv) > 1 else 192
step_gb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
n, L = 1024, synth.NB_FRAME_SAMPLES
ctx = dabgpu.Context(device=0, max_frames=n)
ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts); s = ts.cuda_stream
arena = torch.empty(arena_gb * G, dtype=torch.uint8, device=dev)
base = arena.data_ptr()
base += (-base) % 4096
in_bytes, out_bytes = n * L * 8, n * dabgpu.NB_FRAME_BITS
steps = list(range(0, arena_gb - 2, step_gb))
def t(in_off, out_off, reps=3):
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ctx.mover_frames_dev(base + in_off + synth.NB_NULL * 8, L, n, base + out_off, True, s); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
print("mover ms, 1 024 frames; rows: input window at k GB, columns: output window at k GB + 1.75 GB (inside the same step when k is equal)")
print("      " + " ".join("%5d" % k for k in steps))
vals = []
for ki in steps:
    row = []
    for ko in steps:
        row.append(t(ki * G, ko * G + int(1.75 * G)))
    vals.append(row)
    print("%5d " % ki + " ".join("%5.3f" % v for v in row))
v = np.array(vals)
print("min %.3f  median %.3f  max %.3f   diagonal (same 8 GB step) median %.3f" % (v.min(), np.median(v), v.max(), np.median(np.diag(v))))
