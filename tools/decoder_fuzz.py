#!/usr/bin/env python3
"""Differential fuzz of the two channel decoders on the GPU: the wave-per-codeword kernels (self-checking parallel
traceback, survivors in LDS) against the codeword-per-lane kernels (packed int16, survivors in HBM, serial traceback) --
two independent implementations of the same exact-integer Viterbi -- on inputs chosen to stress the traceback: full-range
noise, small-amplitude noise (many ties), sparse inputs (mostly erasures), saturated values, bursts; FIC and sub-channels
of every protection family, with a carried de-interleaver history of the same kind.  A sample of the codewords is also
held against the CPU oracle.  usage: tools/decoder_fuzz.py [rounds] [frames_per_round]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, dabgpu
from oracle import oracle as O
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
F = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)

def make_soft(kind, shape):
    if kind == "noise":
        return torch.randint(-127, 128, shape, dtype=torch.int8, device=dev, generator=g)
    if kind == "small":
        return torch.randint(-3, 4, shape, dtype=torch.int8, device=dev, generator=g)
    if kind == "sparse":
        x = torch.randint(-127, 128, shape, dtype=torch.int8, device=dev, generator=g)
        keep = torch.rand(shape, device=dev, generator=g) < 0.08
        return torch.where(keep, x, torch.zeros_like(x))
    if kind == "saturated":
        return (torch.randint(0, 2, shape, dtype=torch.int8, device=dev, generator=g) * 254 - 127).to(torch.int8)
    if kind == "bursts":                                       # a clean-looking signal with dead stretches
        x = (torch.randint(0, 2, shape, dtype=torch.int8, device=dev, generator=g) * 160 - 80).to(torch.int8)
        x += torch.randint(-40, 41, shape, dtype=torch.int8, device=dev, generator=g)
        dead = (torch.arange(shape[-1], device=dev) // 400) % 5 == 0
        return torch.where(dead, torch.zeros_like(x), x)
    raise ValueError(kind)

def subchannels(rng):
    """a random non-overlapping set drawn from every family"""
    pool = [dabgpu.subchannel(0, br, level=lv, eep_type=0) for br in (8, 16, 32, 48, 64, 96, 128, 192) for lv in (1, 2, 3, 4)]
    pool += [dabgpu.subchannel(0, br, level=lv, eep_type=1) for br in (32, 64, 96, 128) for lv in (1, 2, 3, 4)]
    pool += [dabgpu.uep_subchannel(int(i), 0) for i in range(64)]
    rng.shuffle(pool)
    out, cu = [], 0
    for sc in pool:
        if cu + sc.length <= 864 and len(out) < 6:
            out.append(dabgpu.Subchannel(cu, sc.length, sc.is_uep, sc.eep_type, sc.protection_level, sc.bitrate_kbps)); cu += sc.length
    return out

wave = dabgpu.Context(device=0, max_frames=F, flags=dabgpu.FLAG_VITERBI_WAVE)
lane = dabgpu.Context(device=0, max_frames=F, flags=dabgpu.FLAG_VITERBI_LANE)
rng = np.random.default_rng(2026)
kinds = ["noise", "small", "sparse", "saturated", "bursts"]
total_cw = total_steps = mism = oracle_cw = oracle_bad = 0
t0 = time.time()
for r in range(rounds):
    g.manual_seed(1000 + r)
    kind = kinds[r % len(kinds)]
    n_streams = int(rng.choice([1, 2, 4])); fps = F // n_streams; n = n_streams * fps
    soft = make_soft(kind, (n, dabgpu.NB_FRAME_BITS))
    scs = subchannels(rng)
    hin = [make_soft(kind, (n_streams, 15, sc.length * 64)) for sc in scs]
    res = []
    for c in (wave, lane):
        fib = torch.zeros((n, 12, 32), dtype=torch.uint8, device=dev); ok = torch.zeros((n, 12), dtype=torch.uint8, device=dev)
        outs = [torch.zeros((n_streams, fps * 4, sc.bitrate_kbps * 3), dtype=torch.uint8, device=dev) for sc in scs]
        hout = [torch.zeros_like(h) for h in hin]
        torch.cuda.synchronize()                               # (inputs and zero-fills run on torch's stream, the decoder on its own)
        c.decode_frames_dev(soft.data_ptr(), dabgpu.NB_FRAME_BITS, n_streams, fps, fib.data_ptr(), ok.data_ptr(), scs,
                            [h.data_ptr() for h in hin], [h.data_ptr() for h in hout], [o.data_ptr() for o in outs], None)
        c.sync()
        res.append((fib, ok, outs, hout))
    bad = int((res[0][0] != res[1][0]).any(dim=2).any(dim=1).sum()) + int((res[0][1] != res[1][1]).any(dim=1).sum())
    for k in range(len(scs)):
        bad += int((res[0][2][k] != res[1][2][k]).any(dim=2).sum()) + int((res[0][3][k] != res[1][3][k]).any())
    mism += bad
    total_cw += n * 4 * (1 + len(scs))
    total_steps += n * 4 * (774 + sum(sc.bitrate_kbps * 24 + 6 for sc in scs))
    # the oracle on a few codewords of this round: two FIC frames and one logical frame of the first sub-channel
    for f in (0, n - 1):
        ofib, ook = O.fic_decode(soft[f, :9216].cpu().numpy())
        oracle_cw += 4; oracle_bad += int((res[0][0][f].cpu().numpy() != ofib).any() or (res[0][1][f].cpu().numpy() != ook).any())
    sc = scs[0]
    if sc.is_uep:
        idx = [i for i in range(64) if dabgpu.uep_subchannel(i, 0).bitrate_kbps == sc.bitrate_kbps and dabgpu.uep_subchannel(i, 0).protection_level == sc.protection_level][0]
        mask, kept, nsteps, _ = O.uep_puncture_mask(idx)
    else:
        mask, kept, nsteps, _ = O.eep_puncture_mask(sc.eep_type, sc.protection_level, sc.bitrate_kbps)
    cifs = soft[:fps, dabgpu.NB_FIC_BITS:].reshape(fps * 4, 55296)[:, sc.start_address * 64:(sc.start_address + sc.length) * 64].cpu().numpy()
    padded = np.concatenate([hin[0][0].cpu().numpy(), cifs])
    for t in (0, 7, fps * 4 - 1):
        want = O.msc_decode_lf(O.time_deinterleave(padded[t:t + 16])[:kept], mask, nsteps)
        oracle_cw += 1; oracle_bad += int((res[0][2][0][0, t].cpu().numpy() != want).any())
    print("round %2d  %-9s %d streams x %d frames, %d sub-channels (%s): %s" % (
        r, kind, n_streams, fps, len(scs), ", ".join("%s%d k" % ("UEP " if s.is_uep else "EEP %d-%s " % (s.protection_level, "AB"[s.eep_type]), s.bitrate_kbps) for s in scs),
        "identical" if bad == 0 else "%d DIFFERENCES" % bad), flush=True)
print("%d rounds, %d codewords, %.1f G trellis steps through both decoders in %.0f s: %d differences; %d codewords also against the CPU oracle: %d differences" % (
    rounds, total_cw, total_steps / 1e9, time.time() - t0, mism, oracle_cw, oracle_bad))
sys.exit(1 if mism or oracle_bad else 0)
