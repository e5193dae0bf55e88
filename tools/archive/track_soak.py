#!/usr/bin/env python3
"""GPU box: long-run check of the timing tracker.  Two streams whose sample clocks are off by +80 and -60 ppm, NF frames
each (default 4000 = 6.4 minutes of signal), acquired once and then tracked capture by capture (16 frames per call): every
frame must be demodulated exactly once, in order, with all twelve FIBs equal to the transmitted ones, no desync, and the
drift estimate on the mark at the end.  usage: tools/track_soak.py [NF]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, dabgpu
from dabgpu import synth
NF = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
L, NULL, HALF = synth.NB_FRAME_SAMPLES, synth.NB_NULL, 12          # HALF: group delay of synth.resample
dev = torch.device("cuda", 0)
ppms, C = (80.0, -60.0), 16
S = len(ppms)
n_cap, adv, MF = (C + 1) * L + 4096, C * L, C + 2
ens, xs = [], []
for s, ppm in enumerate(ppms):
    e = synth.Ensemble(seed=1900 + s, n_frames=4)
    tx = torch.from_numpy(e.iq().ravel()).to(dev).repeat((NF + 3) // 4)[:NF * L]
    x = synth.resample(tx, ppm)
    n = torch.arange(x.shape[0], device=dev, dtype=torch.float64)
    cfo = (2.3 - 4.1 * s) / 2048.0
    x = x * torch.exp(2j * np.pi * cfo * n).to(torch.complex64)
    del n, tx
    g = torch.Generator(device=dev); g.manual_seed(177 + s)
    sigma = float(np.sqrt(0.5 * 10 ** (-16.0 / 10)))
    x = x + sigma * (torch.randn(x.shape, generator=g, device=dev) + 1j * torch.randn(x.shape, generator=g, device=dev))
    ens.append(e); xs.append(x[50000 + 12345 * s:])
n_total = min(int(x.shape[0]) for x in xs)
d_x = torch.stack([x[:n_total] for x in xs]).contiguous()
del xs
c = dabgpu.Context(0, 64)
c.streams_reset(S)
frames = torch.zeros((S, MF, 32), dtype=torch.uint8, device=dev)
counts = torch.zeros(S, dtype=torch.int32, device=dev)
soft = torch.zeros((S * MF, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev)
fib = torch.zeros((S * MF, 12, 32), dtype=torch.uint8, device=dev)
ok = torch.zeros((S * MF, 12), dtype=torch.uint8, device=dev)
cfg = dabgpu.track_cfg(auto_acquire=1)
seen = [[] for _ in range(S)]
base, call, bad = 0, 0, 0
t0 = time.perf_counter()
torch.cuda.synchronize()
while base + n_cap <= n_total:
    p = d_x.data_ptr() + base * 8
    c.ofdm_demod_tracked_dev(p, n_total, S, n_cap, MF, adv, soft.data_ptr(), frames.data_ptr(), counts.data_ptr(), cfg=cfg)
    c.fic_decode_dev(soft.data_ptr(), dabgpu.NB_FRAME_BITS, S * MF, fib.data_ptr(), ok.data_ptr())
    c.sync()
    fr = frames.cpu().numpy().view(dabgpu.ACQUIRED_FRAME_DTYPE).reshape(S, MF)
    cnt = counts.cpu().numpy()
    fib_h, ok_h = fib.cpu().numpy().reshape(S, MF, 12, 32), ok.cpu().numpy().reshape(S, MF, 12)
    for s in range(S):
        for j in range(cnt[s]):
            pos = base + int(fr[s, j]["start"]) + 64 + HALF + 50000 + 12345 * s
            k = int(round((pos / (1 + ppms[s] * 1e-6) - NULL) / L))
            good = fr[s, j]["flags"] == 3 and abs(pos / (1 + ppms[s] * 1e-6) - NULL - k * L) <= 1.5 and ok_h[s, j].all() and \
                (fib_h[s, j] == ens[s].fibs[k % 4]).all()
            if not good and bad < 3:
                print("first failures:", call, s, j, int(fr[s, j]["flags"]), pos / (1 + ppms[s] * 1e-6) - NULL - k * L, bool(ok_h[s, j].all()),
                      bool((fib_h[s, j] == ens[s].fibs[k % 4]).all()))
            bad += not good
            seen[s].append(k)
    base += adv
    call += 1
st = [c.get_stats(s) for s in range(S)]
print("tools/track_soak.py: %d frames per stream (%.0f s of signal), %d calls of %d frames, %.1f s wall" %
      (NF, NF * 0.096, call, C, time.perf_counter() - t0))
for s in range(S):
    in_order = seen[s] == list(range(seen[s][0], seen[s][0] + len(seen[s])))
    print("stream %d: %+.0f ppm  frames demodulated %d (each once, in order: %s)  "
          "desyncs %d  drift estimate %.4f samples/frame (true %.4f)  tracking %d" %
          (s, ppms[s], len(seen[s]), in_order, st[s].total_frames_desync, st[s].drift, ppms[s] * 1e-6 * L, st[s].tracking))
print("frames failing any check: %d" % bad)
sys.exit(0 if bad == 0 and all(x.total_frames_desync == 0 for x in st) else 1)
