#!/usr/bin/env python3
"""GPU box: does the fused OFDM kernel's HBM rate depend on where the frames lie?  Same kernel, same bytes, frames
placed at different strides (samples) and cut into different numbers of symbol runs.
usage: tools/ofdm_layout.py [n_frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, dabgpu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device("cuda", 0)
fo = ((torch.rand(n, device=dev) - 0.5) * 0.8 / 2048).float()
soft = torch.empty((n, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev)
cyc = torch.zeros((n, 76, 2), dtype=torch.float32, device=dev)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); s = st.cuda_stream
A = 1782016
def t(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts), float(np.mean(ts))
print("# tools/ofdm_layout.py %d frames, lib %s" % (n, os.environ.get("DABGPU_LIB", "default")))
for stride in (196608, 196608 + 512, 196608 + 2552, 196608 + 4096 + 64, 76 * 2552, 76 * 2552 + 8, 262144):
    iq = torch.randn((n * stride + 4096, 2), dtype=torch.float32, device=dev)
    for parts in (0, 1, 5, 15):
        ctx = dabgpu.Context(0, n, ofdm_symbol_runs=parts)
        mn, av = t(lambda: ctx.ofdm_demod_frames_dev(iq.data_ptr() + 2656 * 8, stride, n, fo.data_ptr(), soft.data_ptr(), cyc.data_ptr(), None, s))
        print("stride %7d samples (%+6d)  runs/frame %2s : min %.3f ms mean %.3f ms  %.0f GB/s algorithmic" % (
            stride, stride - 196608, parts or "auto", mn, av, A * n / mn / 1e6))
        ctx.close()
    del iq
