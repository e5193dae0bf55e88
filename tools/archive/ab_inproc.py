#!/usr/bin/env python3
"""A/B between builds of libdabgpu.so INSIDE one process: same buffers, same physical pages, launches alternated --
the process-to-process spread of the HBM-bound kernels (+-4 %) drops out and 0.3 % differences become visible.
usage: tools/ab_inproc.py <mode> <n_frames> <rounds> <reps> lib1.so lib2.so ...   ("default" = the in-tree library)
modes: ofdm (fused front end, NCO + cyc) | dd (the same with decision-directed sums, no cyclic prefix read) |
       select_dd (dd with a soft-bit selection) | bare (no estimator output at all) | fft (FFT stage only) | select (front end with a soft-bit selection) |
       acquire (null search + PRS sync on unaligned captures) | decode (FIC + one sub-channel, grouped launch) |
       multiplex (FIC + 18 sub-channels) | sync (PRS synchronisation of every frame, fine time only) | sync_coarse (with the +-200 carrier search)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import torch, dabgpu
mode, n, rounds, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
paths = [dabgpu.LIB_PATH if p == "default" else p for p in sys.argv[5:]]
dev = torch.device("cuda", 0)
L, NB = dabgpu.NB_FRAME_SAMPLES, dabgpu.NB_FRAME_BITS
st = torch.cuda.Stream(); torch.cuda.set_stream(st); s = st.cuda_stream
E = 64; F = max(1, n // E)
if mode in ("ofdm", "fft", "select", "acquire", "dd", "select_dd", "bare", "sync", "sync_coarse"):
    iq = torch.empty((n, L, 2), dtype=torch.float32, device=dev).normal_()
    fo = torch.full((n,), 1.0e-4, dtype=torch.float32, device=dev)
if mode in ("ofdm", "select"):
    soft = torch.zeros((n, NB), dtype=torch.int8, device=dev); cyc = torch.zeros((n, 76, 2), dtype=torch.float32, device=dev)
    outs = [soft, cyc]
    sel = dabgpu.soft_selection([dabgpu.subchannel(0, 64, level=3)], True) if mode == "select" else None
    def run(c): c.ofdm_demod_frames_dev(iq.data_ptr(), L, n, fo.data_ptr(), soft.data_ptr(), cyc.data_ptr(), None, s)
elif mode in ("dd", "select_dd", "bare"):
    soft = torch.zeros((n, NB), dtype=torch.int8, device=dev); dd4 = torch.zeros((n, 76, 2), dtype=torch.float32, device=dev)
    outs = [soft, dd4]
    sel = dabgpu.soft_selection([dabgpu.subchannel(0, 64, level=3)], True) if mode == "select_dd" else None
    if mode == "bare":
        def run(c): c.ofdm_demod_frames_dev(iq.data_ptr(), L, n, fo.data_ptr(), soft.data_ptr(), None, None, s)
    else:
        def run(c): c.ofdm_demod_frames_dd_dev(iq.data_ptr(), L, n, fo.data_ptr(), soft.data_ptr(), dd4.data_ptr(), s)
elif mode in ("sync", "sync_coarse"):
    sres = torch.zeros((n, 4), dtype=torch.int32, device=dev); outs = [sres]; sel = None
    def run(c): c.sync_prs_dev(iq.data_ptr() + 2656 * 8, L, n, fo.data_ptr(), 200 if mode == "sync_coarse" else 0, sres.data_ptr(), s)
elif mode == "fft":
    spec = torch.zeros((n, 76, 2048, 2), dtype=torch.float32, device=dev); outs = [spec]; sel = None
    def run(c): c.fft_symbols_dev(iq.data_ptr(), L, n, fo.data_ptr(), spec.data_ptr(), s)
elif mode == "acquire":
    # E captures of F frames each; a null symbol's worth of the start is cut so that frames sit at odd places
    for f in range(n): iq[f, :2656] *= 0.02
    cap = iq.view(E, F * L, 2)[:, 45825:, :].contiguous(); ns = cap.shape[1]; del iq
    frames = torch.zeros((E, F, 8), dtype=torch.int32, device=dev); counts = torch.zeros((E,), dtype=torch.int32, device=dev)
    outs = [frames, counts]; sel = None
    def run(c): c.acquire_dev(cap.data_ptr(), ns, E, ns, F, frames.data_ptr(), counts.data_ptr(), None, s)
else:
    soft = torch.randint(-127, 128, (n, NB), dtype=torch.int8, device=dev)
    fib = torch.zeros((n, 12, 32), dtype=torch.uint8, device=dev); crc = torch.zeros((n, 12), dtype=torch.uint8, device=dev)
    scs, cu = [], 0
    if mode == "decode":
        scs = [dabgpu.subchannel(0, 64, level=3)]
    else:
        for br, lvl, k in ((64, 3, 10), (48, 3, 4), (32, 2, 3)):
            for _ in range(k):
                x = dabgpu.subchannel(cu, br, level=lvl); scs.append(x); cu += x.length
        scs.append(dabgpu.uep_subchannel(35, cu))
    mo = [torch.zeros((E, F * 4, x.bitrate_kbps * 3), dtype=torch.uint8, device=dev) for x in scs]
    outs = [fib, crc] + mo; sel = None
    def run(c): c.decode_frames_dev(soft.data_ptr(), NB, E, F, fib.data_ptr(), crc.data_ptr(), scs, None, None, [o.data_ptr() for o in mo], s)
libs = []
for p in paths:
    c = dabgpu.Context(0, n, library=dabgpu.load_library(p))
    if sel is not None: c.set_soft_selection(sel)
    libs.append((os.path.basename(p), c))
def checksum():
    return tuple(int(o.contiguous().view(torch.uint8).to(torch.int64).sum().item()) for o in outs)
ref = None
for name, c in libs:                            # warm up; the builds must agree
    for o in outs: o.zero_()
    for _ in range(2): run(c)
    torch.cuda.synchronize()
    chk = checksum()
    if ref is None: ref = chk
    print("# %-28s checksum %s%s" % (name, chk, "" if chk == ref else "   <-- DIFFERS"))
res = {name: [] for name, _ in libs}
for r in range(rounds):
    for name, c in libs:
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): run(c)
        e1.record(); torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / reps)
for name in res:
    v = res[name]
    print("%-10s %-28s mean %.3f ms  min %.3f  | %s" % (mode, name, sum(v) / len(v), min(v), " ".join("%.3f" % x for x in v)))
for _, c in libs: c.close()
