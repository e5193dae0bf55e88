"""Randomised differential test of the channel decoder: random protection profiles (EEP-A, EEP-B, UEP), shapes
(streams x frames, ragged group tails, whole-group and straddling streams), positions in the CIF, with and without
carried history -- the three decoder formulations (one wave per codeword; one lane per codeword with the prep
kernel or the fused forward pass; the grouped launch) must agree byte for byte, and agree with the oracle on
sampled logical frames."""
import numpy as np
import pytest
import torch

import dabgpu
from conftest import make_ctx
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def random_subchannel(rng, used):
    for _ in range(100):
        kind = rng.integers(0, 3)
        if kind == 0:
            br, lvl = int(rng.choice([8, 16, 24, 32, 48, 56, 64, 72, 96, 128, 160])), int(rng.integers(1, 5))
            sc = dabgpu.subchannel(0, br, level=lvl, eep_type=0)
            mask, kept, nsteps, _ = O.eep_puncture_mask(0, lvl, br)
        elif kind == 1:
            br, lvl = int(rng.choice([32, 64, 96, 128])), int(rng.integers(1, 5))
            sc = dabgpu.subchannel(0, br, level=lvl, eep_type=1)
            mask, kept, nsteps, _ = O.eep_puncture_mask(1, lvl, br)
        else:
            idx = int(rng.integers(0, 58))
            sc = dabgpu.uep_subchannel(idx, 0)
            mask, kept, nsteps, _ = O.uep_puncture_mask(idx)
        start = int(rng.integers(0, 864 - sc.length + 1))
        if not used[start:start + sc.length].any():
            used[start:start + sc.length] = True
            sc.start_address = start
            return sc, mask, kept, nsteps
    return None


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("DAB_RANDOM_SEEDS", "12"))))
def test_decoder_formulations_agree(seed):
    rng = np.random.default_rng(1000 + seed)
    n_streams = int(rng.integers(1, 5))
    fps = int(rng.choice([1, 3, 5, 16, 16, 32, 17]))          # CIFs per stream: multiples of 64 and not
    n = n_streams * fps
    used = np.zeros(864, bool)
    subs = [s for s in (random_subchannel(rng, used) for _ in range(int(rng.integers(1, 5)))) if s]
    with_hist = bool(rng.integers(0, 2))
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(seed)
    soft = torch.randint(-127, 128, (n, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev, generator=g)
    if seed % 3 == 0:
        soft[:, ::7] = 0                                      # erasures sprinkled in
    hin = [torch.randint(-127, 128, (n_streams, 15, sc.length * 64), dtype=torch.int8, device=dev, generator=g)
           for sc, *_ in subs]

    def run(mode, grouped):
        c = make_ctx(mode, max_frames=8)
        fib = torch.zeros((n, 12, 32), dtype=torch.uint8, device=dev); ok = torch.zeros((n, 12), dtype=torch.uint8, device=dev)
        outs = [torch.zeros((n_streams, fps * 4, sc.bitrate_kbps * 3), dtype=torch.uint8, device=dev) for sc, *_ in subs]
        houts = [torch.zeros_like(h) for h in hin]
        hi = [h.data_ptr() for h in hin] if with_hist else None
        if grouped:
            c.decode_frames_dev(soft.data_ptr(), dabgpu.NB_FRAME_BITS, n_streams, fps, fib.data_ptr(), ok.data_ptr(),
                                [s[0] for s in subs], hi, [h.data_ptr() for h in houts], [o.data_ptr() for o in outs], None)
        else:
            c.fic_decode_dev(soft.data_ptr(), dabgpu.NB_FRAME_BITS, n, fib.data_ptr(), ok.data_ptr(), None)
            for i, (sc, *_r) in enumerate(subs):
                c.msc_decode_dev(sc, soft.data_ptr(), dabgpu.NB_FRAME_BITS, n_streams, fps, hin[i].data_ptr() if with_hist else None,
                                 houts[i].data_ptr(), outs[i].data_ptr(), None)
        c.sync()
        res = (fib.cpu().numpy(), ok.cpu().numpy(), [o.cpu().numpy() for o in outs], [h.cpu().numpy() for h in houts])
        c.close()
        return res

    ref = run(0, False)
    for mode, grouped in ((1, False), (1, True), (None, True)):
        got = run(mode, grouped)
        assert (got[0] == ref[0]).all() and (got[1] == ref[1]).all(), (mode, grouped)
        for i in range(len(subs)):
            assert (got[2][i] == ref[2][i]).all(), (mode, grouped, i)
            assert (got[3][i] == ref[3][i]).all(), (mode, grouped, i)
    # oracle on a few logical frames of every sub-channel and one FIC
    soft_h = soft.cpu().numpy()
    ofib, ook = O.fic_decode(soft_h[n - 1, :9216])
    assert (ref[0][n - 1] == np.asarray(ofib).reshape(12, 32)).all() and (ref[1][n - 1] == ook).all()
    for i, (sc, mask, kept, nsteps) in enumerate(subs):
        nbits = sc.length * 64
        s = int(rng.integers(0, n_streams))
        cifs = soft_h[s * fps:(s + 1) * fps, 9216:].reshape(fps * 4, 55296)[:, sc.start_address * 64:sc.start_address * 64 + nbits]
        h = hin[i][s].cpu().numpy() if with_hist else np.zeros((15, nbits), np.int8)
        padded = np.concatenate([h, cifs])
        for t in {0, fps * 4 - 1, int(rng.integers(0, fps * 4))}:
            want = O.msc_decode_lf(O.time_deinterleave(padded[t:t + 16])[:kept], mask, nsteps)
            assert (ref[2][i][s, t] == want).all(), (i, s, t)
