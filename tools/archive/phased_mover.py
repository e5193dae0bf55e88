#!/usr/bin/env python3
"""GPU box: does the front end's 13 % write stream cost less when the whole chip writes in bursts?

The library's geometry mover (dabgpu_mover_frames_dev: the fused front end's loads, stores and occupancy without its arithmetic)
on the bench's buffers (64 x 256 frames, the placed IQ / soft-bit pair and a plain pair), plain and with
DABGPU_MOVER_BURSTS=period_us,window_percent,hold (csrc/ofdm_kernels.hip: output held back for up to `hold` symbols and
stored only inside a chip-wide clock window; no loads start inside the window).  One process, alternating launches.

The DABGPU_MOVER_BURSTS hook and its two kernels exist in the library of commit 0791219 only (the experiment was rejected,
profiles/r05_write_bursts.md, and the product library reads no environment); check that commit out and `make -C
sdrplusplus-dab-radio-plugin_amd/csrc` to run this again.  tools/ubench/phased_rw.hip shows the effect without the library.

usage: python3 tools/phased_mover.py [--frames 16384] > gpurun_out/phased_mover.txt
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sdrplusplus-dab-radio-plugin_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16384)
    ap.add_argument("--reps", type=int, default=4)
    args = ap.parse_args()
    import torch
    import dabgpu
    from dabgpu import synth
    dev = torch.device("cuda", 0)
    n, L = args.frames, synth.NB_FRAME_SAMPLES
    ctx = dabgpu.Context(device=0, max_frames=n)
    ts = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(ts)
    stream = ts.cuda_stream
    A = (76 * 2552 * 8 + 230400) * n

    def timed(d_iq, d_soft, env):
        if env:
            os.environ["DABGPU_MOVER_BURSTS"] = env
        else:
            os.environ.pop("DABGPU_MOVER_BURSTS", None)
        ev = []
        for i in range(1 + args.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ctx.mover_frames_dev(d_iq, L, n, d_soft, True, stream)
            e1.record()
            if i:
                ev.append((e0, e1))
        torch.cuda.synchronize()
        os.environ.pop("DABGPU_MOVER_BURSTS", None)
        return min(a.elapsed_time(b) for a, b in ev)

    pairs = []
    d_iq, d_soft, rep = ctx.alloc_frame_buffers(n, L, dabgpu.PLACE_DOMAINS)
    pairs.append(("placed pair (domains seen %d, pair / one domain %.3f)" % (rep.n_domains, rep.pair_over_same_domain)
                  if rep.method == 1 else "placement fell back to a plain pair", d_iq + synth.NB_NULL * 8, d_soft))
    iq2 = torch.zeros((n, L), dtype=torch.complex64, device=dev)
    soft2 = torch.zeros((n, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev)
    pairs.append(("plain pair (two torch allocations)", iq2.data_ptr() + synth.NB_NULL * 8, soft2.data_ptr()))
    grid = [(p, w, h) for h in (2, 3, 4) for p in (12, 16, 20, 24, 28, 32, 40) for w in (10, 15, 20, 25)]
    grid += [(p, w, 1) for p in (8, 10, 12, 14) for w in (15, 25)] + [(p, w, 8) for p in (40, 60) for w in (15, 25)]
    for name, a_iq, a_soft in pairs:
        base = timed(a_iq, a_soft, None)
        print("%s: %d frames, %.2f GB per launch; plain mover %.3f ms = %.0f GB/s = %.3f of 8 TB/s" % (name, n, A / 1e9, base, A / base / 1e6,
                                                                                                       A / base / 1e6 / 8000))
        rows = []
        for p, w, h in grid:
            t = timed(a_iq, a_soft, "%g,%g,%d" % (p, w, h))
            rows.append((t, p, w, h))
        print("  stores left dirty in L2, written back (buffer_wbl2) by one wave in `every` when the window opens (period_us window_% every gate_loads ms):")
        for p, w in ((4, 25), (8, 25), (12, 25), (20, 25), (40, 25)):
            for every in (1, 4, 16, 64, 512):
                for gate in (0, 1):
                    t = timed(a_iq, a_soft, "%g,%g,0,%d,%d" % (p, w, every, gate))
                    print("    %5.1f %4.1f %4d %d %.3f (%+.1f %%)" % (p, w, every, gate, t, (t / base - 1) * 100))
        again = timed(a_iq, a_soft, None)
        for h in sorted({r[3] for r in rows}):
            best = sorted(r for r in rows if r[3] == h)[:4]
            print("  hold %d: " % h + "; ".join("%.3f ms (%+.1f %%) at %g us / %g %%" % (t, (t / base - 1) * 100, p, w) for t, p, w, _ in best))
        print("  plain again %.3f ms" % again)
        print("  all rows (hold period_us window_%% ms):")
        for t, p, w, h in rows:
            print("    %d %5.1f %4.1f %.3f" % (h, p, w, t))


if __name__ == "__main__":
    main()
