#!/usr/bin/env python3
"""Workload for the HBM-traffic PMC passes: one calibration copy of known size, then the fused OFDM kernel on
the bench shape (n frames of random IQ).  Run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import torch, dabgpu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0)
cal = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev).normal_()     # 1 GiB
for _ in range(3):
    cal2 = cal.clone()                                                             # reads 1 GiB, writes 1 GiB
torch.cuda.synchronize()
iq = torch.randn((n, 196608, 2), dtype=torch.float32, device=dev)
fo = ((torch.rand(n, device=dev) - 0.5) * 0.8 / 2048).float()
soft = torch.empty((n, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev)
cyc = torch.zeros((n, 76, 2), dtype=torch.float32, device=dev)
ctx = dabgpu.Context(0, n); st = torch.cuda.Stream(); torch.cuda.set_stream(st)
mode = sys.argv[2] if len(sys.argv) > 2 else "cp"
for _ in range(5):
    if mode == "dd":      # the bench's own-estimator leg: decision-directed sums, no cyclic prefix read
        ctx.ofdm_demod_frames_dd_dev(iq.data_ptr() + 2656 * 8, 196608, n, fo.data_ptr(), soft.data_ptr(), cyc.data_ptr(), st.cuda_stream)
    else:                 # what the bench's timed step launches: the cyclic-prefix correlations (the reference's data flow)
        ctx.ofdm_demod_frames_dev(iq.data_ptr() + 2656 * 8, 196608, n, fo.data_ptr(), soft.data_ptr(), cyc.data_ptr(), None, st.cuda_stream)
torch.cuda.synchronize()
