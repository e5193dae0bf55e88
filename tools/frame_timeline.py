#!/usr/bin/env python3
"""The plugin's one-frame path under the kernel tracer: `rocprofv3 --kernel-trace -- python3 tools/frame_timeline.py work`
runs 3 x 32 frames through dabgpu_ofdm_demod_stream_frame -> dabgpu_decode_stream_frames (page-locked buffers, FIC + one
64 kbit/s sub-channel) and prints each call's host wall clock; `tools/frame_timeline.py report <trace.csv>` then lays the
kernels of the calls end to end: how long each ran, and how long the device sat between them -- what a call's wall clock
is made of besides its kernels."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)


def work():
    import ctypes as C
    import numpy as np, dabgpu
    if os.environ.get("WITH_TORCH"):                          # (the bench's process has torch's runtime state in it: does that matter?)
        import torch
        torch.zeros(16, device="cuda").sum().item()
        if os.environ.get("WITH_TORCH") == "big":
            keep = torch.empty(int(20e9), dtype=torch.uint8, device="cuda")
    from dabgpu import synth
    rng = np.random.default_rng(0)
    e = synth.Ensemble(seed=1, n_frames=4)
    n_host, M, used = 32, 64, 76 * 2552
    tx = np.tile(e.iq().ravel(), (n_host + 4) // 4 + 1)
    rx = synth.channel(tx, snr_db=20.0, cfo=0.2 / 2048, rng=rng)
    L = synth.NB_FRAME_SAMPLES
    p_iq = dabgpu.PinnedArray((n_host, used), np.complex64)
    for f in range(n_host):
        lo = f * L + synth.NB_NULL - M
        p_iq.array[f] = rx[lo:lo + used]
    p_soft = dabgpu.PinnedArray((dabgpu.NB_FRAME_BITS,), np.int8)
    p_fib = dabgpu.PinnedArray((1, 12, 32), np.uint8); p_ok = dabgpu.PinnedArray((1, 12), np.uint8); p_out = dabgpu.PinnedArray((4, 192), np.uint8)
    lib = dabgpu.lib()
    co, cd = dabgpu.Context(device=0, max_frames=1), dabgpu.Context(device=0, max_frames=1)
    co.streams_reset(1)
    cfg = dabgpu.track_cfg(timing_margin=M)
    fres = dabgpu.FrameResult()
    sc = dabgpu.subchannel(0, 64, level=3)
    arr = (dabgpu.Subchannel * 1)(sc)
    outs = (C.c_void_p * 1)(p_out.array.ctypes.data)
    t_o, t_d = [], []
    for rep in range(3):
        for f in range(n_host):
            t0 = time.perf_counter()
            rc = lib.dabgpu_ofdm_demod_stream_frame(co._h, 0, p_iq.array[f].ctypes.data, 1 if (rep == 0 and f == 0) else 0, C.byref(cfg),
                                                    p_soft.array.ctypes.data, None, C.byref(fres))
            t1 = time.perf_counter()
            rc2 = lib.dabgpu_decode_stream_frames(cd._h, p_soft.array.ctypes.data, dabgpu.NB_FRAME_BITS, 1, p_fib.array.ctypes.data,
                                                  p_ok.array.ctypes.data, arr, 1, outs)
            t2 = time.perf_counter()
            assert rc == 0 and rc2 == 0 and fres.flags == 3
            if rep > 0:
                t_o.append(t1 - t0); t_d.append(t2 - t1)
    print("host wall clock per call, us: ofdm_demod_stream_frame mean %.1f min %.1f | decode_stream_frames mean %.1f min %.1f | calls %d"
          % (np.mean(t_o) * 1e6, np.min(t_o) * 1e6, np.mean(t_d) * 1e6, np.min(t_d) * 1e6, len(t_o)))


def report(path):
    import csv
    rows = []
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0].replace("dabk::", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
    rows.sort()
    # a "call" = a run of kernels separated from the next by more than 12 us of idle device
    calls, cur = [], []
    for s, e, k in rows:
        if cur and s - cur[-1][1] > 12000:
            calls.append(cur); cur = []
        cur.append((s, e, k))
    if cur:
        calls.append(cur)
    import collections
    shapes = collections.Counter(tuple(k for _, _, k in c) for c in calls)
    for shape, cnt in shapes.most_common(4):
        sel = [c for c in calls if tuple(k for _, _, k in c) == shape][-40:]
        print("%d calls of %d kernels (last %d averaged):" % (cnt, len(shape), len(sel)))
        tot_k = tot_g = 0.0
        for i, k in enumerate(shape):
            d = sum(c[i][1] - c[i][0] for c in sel) / len(sel) / 1e3
            g = sum(c[i][0] - c[i - 1][1] for c in sel) / len(sel) / 1e3 if i else 0.0
            tot_k += d; tot_g += g
            print("   %-34s %7.1f us   (device idle before it: %5.1f us)" % (k, d, g))
        span = sum(c[-1][1] - c[0][0] for c in sel) / len(sel) / 1e3
        print("   first kernel start -> last kernel end %.1f us = kernels %.1f + gaps %.1f" % (span, tot_k, tot_g))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "report":
        report(sys.argv[2])
    else:
        work()
