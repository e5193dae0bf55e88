#!/usr/bin/env python3
"""GPU box: does the HBM domain of the soft-bit buffer still matter once the cyclic prefixes are not read?  One IQ
buffer (16 384 frames of noise), twelve soft-bit buffers allocated one after the other; the front end timed on every
pair in both modes (with the cyclic-prefix correlations / decision-directed), launches alternated, min of 4."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import torch, dabgpu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device("cuda", 0)
L, NB = dabgpu.NB_FRAME_SAMPLES, dabgpu.NB_FRAME_BITS
st = torch.cuda.Stream(); torch.cuda.set_stream(st); s = st.cuda_stream
iq = torch.empty((n, L, 2), dtype=torch.float32, device=dev).normal_()
fo = torch.full((n,), 1.0e-4, dtype=torch.float32, device=dev)
cyc = torch.zeros((n, 76, 2), dtype=torch.float32, device=dev)
softs = [torch.zeros((n, NB), dtype=torch.int8, device=dev) for _ in range(12)]
c = dabgpu.Context(0, n)


def t(fn):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


print("# soft buffer   with prefixes (ms)   decision-directed (ms)")
rows = []
for k, so in enumerate(softs):
    a = t(lambda: c.ofdm_demod_frames_dev(iq.data_ptr() + 2656 * 8, L, n, fo.data_ptr(), so.data_ptr(), cyc.data_ptr(), None, s))
    b = t(lambda: c.ofdm_demod_frames_dd_dev(iq.data_ptr() + 2656 * 8, L, n, fo.data_ptr(), so.data_ptr(), cyc.data_ptr(), s))
    rows.append((a, b))
    print("%2d (+%5.1f GB)   %.3f   %.3f" % (k, (so.data_ptr() - softs[0].data_ptr()) / 1e9, a, b))
# the pair the library's domain-aware arena hands out (physical chunks mapped through the virtual-memory API), same samples
d_iq, d_soft, rep = c.alloc_frame_buffers_placed(n, L)
piq = dabgpu.device_tensor(torch, d_iq, (n, L, 2), torch.float32, dev)
psoft = dabgpu.device_tensor(torch, d_soft, (n, NB), torch.int8, dev)
piq.copy_(iq); torch.cuda.synchronize()
for name, i_ptr, o_ptr in (("placed IQ -> placed soft", d_iq, d_soft), ("placed IQ -> plain soft 0", d_iq, softs[0].data_ptr()),
                           ("placed IQ -> plain soft 1", d_iq, softs[1].data_ptr()), ("plain IQ  -> placed soft", iq.data_ptr(), d_soft)):
    a = t(lambda: c.ofdm_demod_frames_dev(i_ptr + 2656 * 8, L, n, fo.data_ptr(), o_ptr, cyc.data_ptr(), None, s))
    b = t(lambda: c.ofdm_demod_frames_dd_dev(i_ptr + 2656 * 8, L, n, fo.data_ptr(), o_ptr, cyc.data_ptr(), s))
    print("%-28s %.3f   %.3f" % (name, a, b))
print("arena: all chunks %s  iq chunks %s  soft chunks %s  conflicts per mille %d" % (rep.domains.decode(), rep.iq_map.decode(),
      rep.soft_map.decode(), rep.conflicts))
A = [r[0] for r in rows]; B = [r[1] for r in rows]
print("with prefixes: %.3f .. %.3f ms (spread %.1f %%)   decision-directed: %.3f .. %.3f ms (spread %.1f %%)" %
      (min(A), max(A), 100 * (max(A) / min(A) - 1), min(B), max(B), 100 * (max(B) / min(B) - 1)))
import numpy as np
print("correlation between the two columns: %.2f" % np.corrcoef(A, B)[0, 1])
