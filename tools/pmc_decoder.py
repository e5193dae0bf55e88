#!/usr/bin/env python3
"""Workload for the HBM-traffic PMC passes over the channel decoder: a 1 GiB calibration copy, then the
bench's decode call -- dabgpu_decode_frames_dev, FIC + one 64 kbit/s EEP 3-A sub-channel of 64 streams x 256 frames, noise
soft bits, history carried -- five times.  Run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (tools/pmc_decoder.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import torch, dabgpu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device("cuda", 0)
cal = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev).normal_()     # 1 GiB
for _ in range(3):
    cal2 = cal.clone()
torch.cuda.synchronize()
del cal, cal2
E = 64; F = n // E
soft = torch.randint(-127, 128, (n, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev)
fib = torch.zeros((n, 12, 32), dtype=torch.uint8, device=dev); crc = torch.zeros((n, 12), dtype=torch.uint8, device=dev)
sc = dabgpu.subchannel(0, 64, level=3)
msc = torch.zeros((E, F * 4, 192), dtype=torch.uint8, device=dev)
hist = [torch.randint(-127, 128, (E, 15, sc.length * 64), dtype=torch.int8, device=dev) for _ in range(2)]
ctx = dabgpu.Context(0, n); st = torch.cuda.Stream(); torch.cuda.set_stream(st)
torch.cuda.synchronize()
for k in range(5):
    ctx.decode_frames_dev(soft.data_ptr(), dabgpu.NB_FRAME_BITS, E, F, fib.data_ptr(), crc.data_ptr(), [sc], [hist[k & 1].data_ptr()],
                          [hist[(k & 1) ^ 1].data_ptr()], [msc.data_ptr()], st.cuda_stream)
torch.cuda.synchronize()
