"""Randomised differential test of the front end: random SNR / frequency offset / frame start (even and odd sample
offsets) / soft-bit selection / symbol-run split -- every soft bit the kernel writes is within 1 LSB of the oracle's
demodulation of the same samples with the same correction, and nothing outside a selection is touched."""
import os

import numpy as np
import pytest
import torch

import dabgpu
from conftest import make_ctx
from dabgpu import synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu
SYMS = 76 * 2552


@pytest.mark.parametrize("seed", range(int(os.environ.get("DAB_RANDOM_SEEDS", "8"))))
def test_front_end_random(seed):
    rng = np.random.default_rng(7000 + seed)
    parts = int(rng.choice([0, 1, 2, 5, 25, 75]))
    c = make_ctx(None, 8, ofdm_symbol_runs=parts)
    n_frames = 3
    e = synth.Ensemble(seed=seed, n_frames=n_frames)
    snr = None if seed % 4 == 0 else float(rng.uniform(2.0, 30.0))
    cfo = float(rng.uniform(-40.0, 40.0)) / 2048.0
    iq = synth.channel(e.iq().ravel(), snr_db=snr, cfo=cfo, rng=rng).astype(np.complex64)
    pad = int(rng.integers(0, 9))                            # shifts every frame start; odd values -> 8-byte aligned
    buf = np.concatenate([np.zeros(pad, np.complex64), iq, np.zeros(16, np.complex64)])
    frames = np.zeros(4, dabgpu.ACQUIRED_FRAME_DTYPE)
    for f in range(n_frames):
        frames[f]["start"] = pad + f * synth.NB_FRAME_SAMPLES + synth.NB_NULL
        frames[f]["freq_offset"] = np.float32(-cfo + rng.uniform(-0.02, 0.02) / 2048.0)
        frames[f]["flags"] = 3
    frames[3]["start"], frames[3]["flags"] = 5, 1            # an entry that does not fit: must come out erased
    sel = None
    if seed % 2:
        k = int(rng.integers(1, 6))
        firsts = sorted(int(v) * 16 for v in rng.choice(230400 // 16 - 64, k, replace=False))
        sel = [(f0, int(rng.integers(1, 64)) * 16) for f0 in firsts]
        c.set_soft_selection(sel)
    dev = torch.device("cuda", 0)
    d_iq = torch.from_numpy(buf).to(dev)
    d_fr = torch.from_numpy(frames.view(np.uint8).reshape(4, 32)).to(dev)
    d_soft = torch.full((4, dabgpu.NB_FRAME_BITS), 55, dtype=torch.int8, device=dev)
    d_cyc = torch.zeros((4, 76, 2), dtype=torch.float32, device=dev)
    c.ofdm_demod_acquired_dev(d_iq.data_ptr(), buf.size, 1, 4, d_fr.data_ptr(), d_soft.data_ptr(), d_cyc.data_ptr())
    c.sync()
    got = d_soft.cpu().numpy()
    cyc = d_cyc.cpu().numpy().view(np.complex64).reshape(4, 76)
    keep = np.zeros(230400, bool)
    if sel is None:
        keep[:] = True
    else:
        for f0, cnt in sel:
            keep[f0:f0 + cnt] = True
    for f in range(n_frames):
        st = int(frames[f]["start"])
        osoft, _, ocyc, _ = O.ofdm_demod_frame(buf[st:st + SYMS], float(frames[f]["freq_offset"]), want_cyc=True)
        assert np.abs(got[f].astype(np.int32) - osoft.astype(np.int32))[keep].max() <= 1
        assert (got[f][~keep] == 55).all()
        assert np.abs(cyc[f] - ocyc).max() <= 2e-3 * np.abs(ocyc).max()
    assert not got[3].any()
    c.close()
