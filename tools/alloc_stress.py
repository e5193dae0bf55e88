#!/usr/bin/env python3
"""GPU box: dabgpu_alloc_frame_buffers(PLACE_DOMAINS) + free, N rounds in one process (default 300), optionally beside a
child process that holds `hold_gb` of device memory for the duration (the situation round 3's allocator crashed in).
Prints per round: chunks taken, the allocator's own check (mover on the pair / mover in one domain), conflicts per mille,
set-up peak over the pair's size; "done" and the drift of free device memory at the end.
usage: tools/alloc_stress.py [rounds] [hold_gb]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
HOLD = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
holder = None
if HOLD > 0:        # started before this process touches the GPU
    holder = subprocess.Popen([sys.executable, "-c",
                               "import torch,sys,time\n"
                               "t=[torch.empty(1<<30,dtype=torch.uint8,device='cuda') for _ in range(int(%f))]\n"
                               "torch.cuda.synchronize(); print('holding',len(t),'GiB',flush=True)\n"
                               "sys.stdin.read()" % HOLD], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    print(holder.stdout.readline().strip(), flush=True)
import torch, dabgpu
torch.cuda.synchronize()
c = dabgpu.Context(0, 8)
L = dabgpu.NB_FRAME_SAMPLES
final = 3000 * (L * 8 + dabgpu.NB_FRAME_BITS)
d_iq, d_soft, rep = c.alloc_frame_buffers(3000, L, dabgpu.PLACE_DOMAINS)          # once, so that pools are warm
assert rep.method == 1, (rep.fallback_reason, rep.runtime_error)
c.free_frame_buffers(d_iq, d_soft)
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
plain = 0
t0 = time.time()
for k in range(N):
    d_iq, d_soft, rep = c.alloc_frame_buffers(3000, L, dabgpu.PLACE_DOMAINS)
    assert d_iq and d_soft and 0 <= rep.conflicts <= 1000 and rep.setup_peak_bytes <= 1.5 * final
    plain += rep.method == 0
    if k % 25 == 0 or rep.method == 0:
        print(k, rep.method, rep.fallback_reason, rep.runtime_error, rep.n_chunks, "%.3f" % rep.pair_over_same_domain, rep.conflicts,
              "%.2f" % (rep.setup_peak_bytes / final), rep.domains.decode(), flush=True)
    c.free_frame_buffers(d_iq, d_soft)
torch.cuda.synchronize()
print("done: %d rounds in %.1f s, %d plain fallbacks, free memory moved by %d MiB" %
      (N, time.time() - t0, plain, (free0 - torch.cuda.mem_get_info()[0]) >> 20))
c.close()
if holder:
    holder.stdin.close()
    holder.wait()
