#!/usr/bin/env python3
"""GPU box: does the channel decoder (VALU-bound) hide under the front end (HBM-bound) when the batch is cut into
G groups of streams and group g's decode runs on a second stream while group g+1 is demodulated?
usage: tools/overlap_time.py [ensembles] [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, dabgpu
E = int(sys.argv[1]) if len(sys.argv) > 1 else 64
F = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n = E * F
dev = torch.device("cuda", 0)
iq = torch.randn((E, F * 196608, 2), dtype=torch.float32, device=dev)
soft = torch.empty((n, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev)
cyc = torch.zeros((n, 76, 2), dtype=torch.float32, device=dev)
fib = torch.zeros((n, 12, 32), dtype=torch.uint8, device=dev); ok = torch.zeros((n, 12), dtype=torch.uint8, device=dev)
sc = dabgpu.subchannel(0, 64, level=3)
msc = torch.zeros((E, F * 4, 192), dtype=torch.uint8, device=dev)
hist = [torch.zeros((E, 15, sc.length * 64), dtype=torch.int8, device=dev) for _ in range(2)]
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
L = 196608


def run(G, reps=6):
    Eg = E // G
    octx = [dabgpu.Context(0, n) for _ in range(G)]
    for c in octx: c.streams_reset(Eg)
    dctx = dabgpu.Context(0, n)
    def step(k):
        evs = []
        for g in range(G):
            with torch.cuda.stream(sA):
                octx[g].ofdm_demod_streams_dev(iq.data_ptr() + (g * Eg * F * L + 2656) * 8, L, Eg, F, 0.5,
                                               soft.data_ptr() + g * Eg * F * dabgpu.NB_FRAME_BITS,
                                               cyc.data_ptr() + g * Eg * F * 76 * 8, None, sA.cuda_stream)
                e = torch.cuda.Event(); e.record(sA); evs.append(e)
            st = sB if G > 1 else sA
            with torch.cuda.stream(st):
                st.wait_event(evs[g])
                o = g * Eg
                dctx.decode_frames_dev(soft.data_ptr() + o * F * dabgpu.NB_FRAME_BITS, dabgpu.NB_FRAME_BITS, Eg, F,
                                       fib.data_ptr() + o * F * 384, ok.data_ptr() + o * F * 12, [sc],
                                       [hist[k & 1].data_ptr() + o * 15 * sc.length * 64], [hist[(k & 1) ^ 1].data_ptr() + o * 15 * sc.length * 64],
                                       [msc.data_ptr() + o * F * 4 * 192], st.cuda_stream)
        if G > 1:
            e = torch.cuda.Event(); e.record(sB); sA.wait_event(e)      # the next step's front end may reuse `soft` only after this
    for k in range(3): step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(reps): step(3 + k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps * 1e3
    for c in octx: c.close()
    dctx.close()
    return dt


for G in (1, 2, 4, 8, 1, 2, 4):
    print("groups %d: %.3f ms per %d frames -> %.0f frames/s" % (G, run(G), n, n / run(G) * 1e3) if False else "groups %d: %.3f ms per step of %d frames" % (G, run(G), n))
