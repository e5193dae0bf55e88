#!/usr/bin/env python3
"""GPU box: one fresh-process trial of dabgpu_alloc_frame_buffers_placed (3000 frames = 4.4 GiB of samples), in the
state a test process is in (torch initialised, one small allocation made and freed).  Prints the allocator's own check
(mover on the pair / mover inside one domain: ~0.9 apart, ~1.0 together), what it claims, and the chunk map.
Run it in a loop to see how often the classification goes wrong:
    for i in $(seq 80); do python3 tools/placement_trial.py; done | cut -c1-9 | sort | uniq -c"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import torch, dabgpu
torch.cuda.synchronize()
c = dabgpu.Context(0, 8)
d_iq, d_soft, rep = c.alloc_frame_buffers_placed(8, dabgpu.NB_FRAME_SAMPLES)
c.free_frame_buffers(d_iq, d_soft)
torch.cuda.synchronize()
d_iq, d_soft, rep = c.alloc_frame_buffers_placed(3000, dabgpu.NB_FRAME_SAMPLES)
print("%.3f" % rep.pair_over_same_domain, rep.conflicts, rep.n_domains, rep.domains.decode(), rep.iq_map.decode(), rep.soft_map.decode(),
      "%.1f" % rep.classify_ms)
