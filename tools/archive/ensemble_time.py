#!/usr/bin/env python3
"""Whole-ensemble decode: FIC + every sub-channel of a full multiplex (864 CUs) per frame, through
dabgpu_msc_decode_multi_dev.  usage: tools/ensemble_time.py [n_streams] [frames_per_stream]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, dabgpu
E = int(sys.argv[1]) if len(sys.argv) > 1 else 64
F = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n = E * F
dev = torch.device("cuda", 0)
soft = torch.randint(-127, 128, (n, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev)
# a plausible multiplex: 10 x 64k EEP 3-A (48 CU), 4 x 48k EEP 3-A (36), 3 x 32k EEP 2-A (32), one 128k UEP level 3 (96 CU)
scs, cu = [], 0
for br, lvl, k in ((64, 3, 10), (48, 3, 4), (32, 2, 3)):
    for _ in range(k):
        sc = dabgpu.subchannel(cu, br, level=lvl); scs.append(sc); cu += sc.length
scs.append(dabgpu.uep_subchannel(35, cu)); cu += scs[-1].length
assert cu <= 864
outs = [torch.zeros((E, F * 4, sc.bitrate_kbps * 3), dtype=torch.uint8, device=dev) for sc in scs]
hin = [torch.zeros((E, 15, sc.length * 64), dtype=torch.int8, device=dev) for sc in scs]
hout = [torch.zeros_like(h) for h in hin]
fib = torch.zeros((n, 12, 32), dtype=torch.uint8, device=dev); ok = torch.zeros((n, 12), dtype=torch.uint8, device=dev)
ctx = dabgpu.Context(0, 64); st = torch.cuda.Stream(); torch.cuda.set_stream(st); s = st.cuda_stream
def run():
    ctx.fic_decode_dev(soft.data_ptr(), dabgpu.NB_FRAME_BITS, n, fib.data_ptr(), ok.data_ptr(), s)
    ctx.msc_decode_multi_dev(scs, soft.data_ptr(), dabgpu.NB_FRAME_BITS, E, F, [h.data_ptr() for h in hin],
                             [h.data_ptr() for h in hout], [o.data_ptr() for o in outs], s)
for _ in range(2): run()
torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(); reps = 3
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / reps
acs = n * (4 * 774 * 64 + sum(4 * (sc.bitrate_kbps * 24 + 6) * 64 for sc in scs))
print("%d sub-channels, %d CUs, %d frames: %.2f ms -> %.0f frames/s (%.0f x real time), %.2f T ACS/s" % (
    len(scs), cu, n, ms, n / ms * 1e3, n / ms * 1e3 * 0.096, acs / ms / 1e9))
