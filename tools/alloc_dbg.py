import os, sys
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import torch, dabgpu
torch.cuda.synchronize()
c = dabgpu.Context(0, 8)
L = dabgpu.NB_FRAME_SAMPLES
for k in range(12):
    d_iq, d_soft, rep = c.alloc_frame_buffers(3000, L, dabgpu.PLACE_DOMAINS)
    print(k, rep.method, rep.fallback_reason, rep.runtime_error, rep.domains.decode(), rep.iq_map.decode(), rep.soft_map.decode(), hex(d_iq), flush=True)
    c.free_frame_buffers(d_iq, d_soft)
c.close()
