#!/usr/bin/env python3
"""GPU box: dabgpu_alloc_frame_buffers_placed + dabgpu_device_alloc_apart + free, N rounds in one process (default 200).
Prints per round: chunks taken, the allocator's own check (mover on the pair / mover in one domain), conflicts per mille
of both calls; "done" and the drift of free device memory at the end.  usage: tools/alloc_stress.py [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import torch, dabgpu
torch.cuda.synchronize()
c = dabgpu.Context(0, 8)
L = dabgpu.NB_FRAME_SAMPLES
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
d_iq, d_soft, rep = c.alloc_frame_buffers_placed(3000, L)          # once, so that pools are warm
c.free_frame_buffers(d_iq, d_soft)
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
for k in range(N):
    d_iq, d_soft, rep = c.alloc_frame_buffers_placed(3000, L)
    assert rep.method == 1 and d_iq and d_soft and 0 <= rep.conflicts <= 1000
    d_other, ms = c.device_alloc_apart(1 << 30, d_iq, 3000 * L * 8)
    assert d_other and 0.0 <= ms[1] <= 1000.0
    print(k, rep.n_chunks, "%.3f" % rep.pair_over_same_domain, rep.conflicts, "%.0f" % ms[1], flush=True)
    c.device_free(d_other)
    c.free_frame_buffers(d_iq, d_soft)
torch.cuda.synchronize()
print("done, free memory moved by %d MiB" % ((free0 - torch.cuda.mem_get_info()[0]) >> 20))
c.close()
