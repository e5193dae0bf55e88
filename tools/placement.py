#!/usr/bin/env python3
"""Does the time of the HBM-bound front-end kernel depend on WHERE its buffers sit?  One process, several IQ and
soft-bit buffers allocated side by side, the same launch timed on every (IQ buffer, soft buffer) pair.
usage: tools/placement.py [n_frames] [n_buffers]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import torch, dabgpu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
L, NB = dabgpu.NB_FRAME_SAMPLES, dabgpu.NB_FRAME_BITS
st = torch.cuda.Stream(); torch.cuda.set_stream(st); s = st.cuda_stream
iqs = [torch.empty((n, L, 2), dtype=torch.float32, device=dev).normal_() for _ in range(K)]
softs = [torch.zeros((n, NB), dtype=torch.int8, device=dev) for _ in range(K)]
fo = torch.full((n,), 1.0e-4, dtype=torch.float32, device=dev)
cyc = torch.zeros((n, 76, 2), dtype=torch.float32, device=dev)
c = dabgpu.Context(0, n)
def t(iq, soft, reps=4):
    for _ in range(2): c.ofdm_demod_frames_dev(iq.data_ptr(), L, n, fo.data_ptr(), soft.data_ptr(), cyc.data_ptr(), None, s)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): c.ofdm_demod_frames_dev(iq.data_ptr(), L, n, fo.data_ptr(), soft.data_ptr(), cyc.data_ptr(), None, s)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print("IQ buffers at  ", " ".join("%#x" % x.data_ptr() for x in iqs))
print("soft buffers at", " ".join("%#x" % x.data_ptr() for x in softs))
for rnd in range(2):
    print("round %d: rows = IQ buffer, columns = soft buffer, ms per launch" % rnd)
    for i in range(K):
        print("   iq%d  " % i + "  ".join("%.3f" % t(iqs[i], softs[j]) for j in range(K)))

# ---- does the fastest pair stay fast? (same pair re-timed after the other candidates are freed, after more memory is
# allocated, and with other sample values)
best = min(((t(iqs[i], softs[j]), i, j) for i in range(K) for j in range(K)))
_, bi, bj = best
iq, soft = iqs[bi], softs[bj]
print("fastest pair (%d, %d): %.3f ms; again %.3f" % (bi, bj, best[0], t(iq, soft)))
del iqs, softs
torch.cuda.empty_cache()
print("after freeing the other candidates: %.3f" % t(iq, soft))
extra = [torch.zeros((1 << 30,), dtype=torch.uint8, device=dev) for _ in range(8)]
print("after allocating 8 GB more: %.3f" % t(iq, soft))
iq.view(n, L, 2)[:, :2656] *= 0.01
print("with quiet null symbols in the data: %.3f" % t(iq, soft))
iq.zero_()
print("on zeros: %.3f" % t(iq, soft))
cyc2 = torch.zeros((n, 76, 2), dtype=torch.float32, device=dev)
cyc = cyc2
print("with another cyc buffer: %.3f" % t(iq, soft))
c.streams_reset(64)
def t2(reps=4):
    for _ in range(2): c.ofdm_demod_streams_dev(iq.data_ptr(), L, 64, n // 64, 0.9, soft.data_ptr(), cyc.data_ptr(), None, s)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): c.ofdm_demod_streams_dev(iq.data_ptr(), L, 64, n // 64, 0.9, soft.data_ptr(), cyc.data_ptr(), None, s)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print("stream call (state on the device + update kernel): %.3f" % t2())
