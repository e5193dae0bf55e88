#!/usr/bin/env python3
"""Workload for the HBM-traffic PMC passes: one calibration copy of known size, then the fused OFDM kernel on
the bench shape (n frames of random IQ) and -- mode cp, n a multiple of 64 -- the bench's decode call on the soft bits it
left (dabgpu_decode_frames_dev: FIC + one 64 kbit/s EEP 3-A sub-channel of 64 streams, history carried; noise in, so every
survivor path is exercised).  Run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); bench_legs.measure_traffic
and tools/pmc_traffic.sh read the front end's and the decoder's kernels out of the same two passes."""
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
mode = sys.argv[2] if len(sys.argv) > 2 else "cp"
decoder = sys.argv[3] if len(sys.argv) > 3 else "auto"          # bench.py --decoder: which kernels decode (auto: by batch size)
flags = {"auto": 0, "lane": dabgpu.FLAG_VITERBI_LANE, "wave": dabgpu.FLAG_VITERBI_WAVE}[decoder]
ctx = dabgpu.Context(0, n, flags=flags); st = torch.cuda.Stream(); torch.cuda.set_stream(st)
for _ in range(5):
    if mode == "dd":      # the bench's own-estimator leg: decision-directed sums, no cyclic prefix read
        ctx.ofdm_demod_frames_dd_dev(iq.data_ptr() + 2656 * 8, 196608, n, fo.data_ptr(), soft.data_ptr(), cyc.data_ptr(), st.cuda_stream)
    else:                 # what the bench's timed step launches: the cyclic-prefix correlations (the reference's data flow)
        ctx.ofdm_demod_frames_dev(iq.data_ptr() + 2656 * 8, 196608, n, fo.data_ptr(), soft.data_ptr(), cyc.data_ptr(), None, st.cuda_stream)
torch.cuda.synchronize()
if mode == "cp" and n % 64 == 0:
    E = 64 if n % 1024 == 0 else 4; F = n // E                  # (whole groups of 64 codewords per stream: F % 16 == 0)
    fib = torch.zeros((n, 12, 32), dtype=torch.uint8, device=dev); crc = torch.zeros((n, 12), dtype=torch.uint8, device=dev)
    sc = dabgpu.subchannel(0, 64, level=3)
    msc = torch.zeros((E, F * 4, 192), dtype=torch.uint8, device=dev)
    hist = [torch.randint(-127, 128, (E, 15, sc.length * 64), dtype=torch.int8, device=dev) for _ in range(2)]
    for k in range(5):
        ctx.decode_frames_dev(soft.data_ptr(), dabgpu.NB_FRAME_BITS, E, F, fib.data_ptr(), crc.data_ptr(), [sc], [hist[k & 1].data_ptr()],
                              [hist[(k & 1) ^ 1].data_ptr()], [msc.data_ptr()], st.cuda_stream)
    torch.cuda.synchronize()
