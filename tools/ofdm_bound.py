#!/usr/bin/env python3
"""GPU box: what bounds the fused OFDM kernel?  Times the SAME kernel (same instructions, same LDS traffic, same
stores) with its IQ reads coming (a) from HBM as in the bench, (b) from the caches: every frame of the launch is
pointed at the same few frames through dabgpu_ofdm_demod_acquired_dev, so the read stream never leaves L2.
(b) is the kernel's compute + LDS + store time; (a) - (b) is what the HBM read stream costs on top.
usage: tools/ofdm_bound.py [n_frames] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, dabgpu

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
E = 64
F = n // E
dev = torch.device("cuda", 0)
iq = torch.randn((E, F * 196608, 2), dtype=torch.float32, device=dev)
fo = ((torch.rand(n, device=dev) - 0.5) * 0.8 / 2048).float()
soft = torch.empty((n, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev)
cyc = torch.zeros((n, 76, 2), dtype=torch.float32, device=dev)
ctx = dabgpu.Context(0, n); st = torch.cuda.Stream(); torch.cuda.set_stream(st); s = st.cuda_stream
A = 1782016


def acq_table(starts_per_stream, n_streams):
    a = np.zeros((n_streams, len(starts_per_stream)), dabgpu.ACQUIRED_FRAME_DTYPE)
    a["start"] = np.asarray(starts_per_stream, np.int64)[None, :]
    a["freq_offset"] = fo.cpu().numpy().reshape(n_streams, -1)
    a["flags"] = 3
    return torch.from_numpy(a.view(np.uint8).reshape(-1)).to(dev)


def t(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts), float(np.mean(ts))


d_iq = iq.data_ptr()
rows = []
rows.append(("aligned frames from HBM (bench call, NCO + cyc)",
             t(lambda: ctx.ofdm_demod_frames_dev(d_iq + 2656 * 8, 196608, n, fo.data_ptr(), soft.data_ptr(), cyc.data_ptr(), None, s))))
tab_b = acq_table([i * 196608 + 2656 for i in range(F)], E)
rows.append(("acquired-frame call, same frames from HBM",
             t(lambda: ctx.ofdm_demod_acquired_dev(d_iq, F * 196608, E, F, tab_b.data_ptr(), soft.data_ptr(), cyc.data_ptr(), None, s))))
for distinct in (1, 2):
    tab_c = acq_table([(i % distinct) * 196608 + 2656 for i in range(n)], 1)
    rows.append(("every frame reads the same %d frame(s): IQ from L2" % distinct,
                 t(lambda: ctx.ofdm_demod_acquired_dev(d_iq, F * 196608, 1, n, tab_c.data_ptr(), soft.data_ptr(), cyc.data_ptr(), None, s))))
tab_c = acq_table([2656] * n, 1)
rows.append(("same, without the cyclic-prefix correlation output",
             t(lambda: ctx.ofdm_demod_acquired_dev(d_iq, F * 196608, 1, n, tab_c.data_ptr(), soft.data_ptr(), None, None, s))))
rows.append(("aligned frames from HBM, NCO, no cyc (-17 % bytes, -4 % instr.)",
             t(lambda: ctx.ofdm_demod_frames_dev(d_iq + 2656 * 8, 196608, n, fo.data_ptr(), soft.data_ptr(), None, None, s))))
dd4 = torch.zeros((n, 76, 2), dtype=torch.float32, device=dev)
rows.append(("aligned frames from HBM, NCO, decision-directed sums instead of cyc (no prefix read)",
             t(lambda: ctx.ofdm_demod_frames_dd_dev(d_iq + 2656 * 8, 196608, n, fo.data_ptr(), soft.data_ptr(), dd4.data_ptr(), s))))
rows.append(("aligned frames from HBM, no NCO, cyc (same bytes, -14 % instr.)",
             t(lambda: ctx.ofdm_demod_frames_dev(d_iq + 2656 * 8, 196608, n, None, soft.data_ptr(), cyc.data_ptr(), None, s))))
rows.append(("aligned frames from HBM, no NCO, no cyc",
             t(lambda: ctx.ofdm_demod_frames_dev(d_iq + 2656 * 8, 196608, n, None, soft.data_ptr(), None, None, s))))
print("# tools/ofdm_bound.py  %d frames per launch, lib %s" % (n, os.environ.get("DABGPU_LIB", "default")))
for name, (mn, av) in rows:
    a = A if ("cyc" in name and "no cyc" not in name and "instead of cyc" not in name and "without" not in name) else A - 76 * 504 * 8
    print("%-86s min %.3f ms  mean %.3f ms  (%.0f GB/s on the bytes this row moves)" % (name, mn, av, a * n / mn / 1e6))
