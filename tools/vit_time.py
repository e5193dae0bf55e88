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
# the bench's grouped call: FIC + one sub-channel in one launch pair
print("   grouped FIC+1 sub-channel us %.1f" % t(lambda: ctx.decode_frames_dev(soft.data_ptr(), dabgpu.NB_FRAME_BITS, E, F, fib.data_ptr(), crc.data_ptr(), [sc], None, None, [msc.data_ptr()], s)))
# FIC + a full multiplex (18 sub-channels, 856 CUs) in grouped launches
scs, cu = [], 0
for br, lvl, k in ((64, 3, 10), (48, 3, 4), (32, 2, 3)):
    for _ in range(k):
        x = dabgpu.subchannel(cu, br, level=lvl); scs.append(x); cu += x.length
scs.append(dabgpu.uep_subchannel(35, cu))
outs = [torch.zeros((E, F * 4, x.bitrate_kbps * 3), dtype=torch.uint8, device=dev) for x in scs]
print("   grouped FIC+18 sub-channels us %.1f" % t(lambda: ctx.decode_frames_dev(soft.data_ptr(), dabgpu.NB_FRAME_BITS, E, F, fib.data_ptr(), crc.data_ptr(), scs, None, None, [o.data_ptr() for o in outs], s), reps=3))
