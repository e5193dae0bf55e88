#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ from THIS repo's oracle and
synthetic transmitter.  The reference holds no fixtures for this path and its DSP cannot
be built here, so nothing in these files comes from /root/reference.

    python tests/golden/make_golden.py          # everything
    python tests/golden/make_golden.py uep      # only uep_cases.npz (added later; leaves the other files alone)
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd"))
sys.path.insert(0, ROOT)
from dabgpu import synth          # noqa: E402
from oracle import oracle as O    # noqa: E402

PROFILES = [(0, 3, 64), (0, 1, 8), (0, 2, 16), (0, 4, 24), (1, 2, 32), (1, 4, 64)]
UEP_INDICES = [0, 4, 13, 15, 29]      # 32k/5, 32k/1 (4 padding bits), 56k/2 (8), 64k/4 (three runs), 112k/5


def make_uep():
    """Viterbi cases for UEP profiles: clean, noisy, pure noise, saturated/erased -- punctured soft bits in,
    bytes out (no energy dispersal), as the EEP cases in viterbi_cases.npz."""
    rng = np.random.default_rng(0x0E9)
    out = {}
    for idx in UEP_INDICES:
        mask, kept, nsteps, _ = O.uep_puncture_mask(idx)
        bits = rng.integers(0, 2, nsteps - 6, dtype=np.uint8)
        tx = np.where(O.conv_encode(bits)[mask.astype(bool)] > 0, 100.0, -100.0)
        cw = np.stack([
            np.clip(tx, -127, 127),
            np.clip(tx + rng.normal(0, 90, kept), -127, 127),
            rng.integers(-127, 128, kept).astype(np.float64),
            rng.choice([-127.0, 127.0, 0.0], kept),
        ]).astype(np.int8)
        out["uep%d_punct" % idx] = cw
        out["uep%d_bytes" % idx] = np.stack([np.packbits(O.viterbi(O.depuncture(c, mask))) for c in cw])
        out["uep%d_truth" % idx] = np.packbits(bits)
    np.savez_compressed(os.path.join(HERE, "uep_cases.npz"), **out)


def main():
    rng = np.random.default_rng(0x601D)
    # --- FIC: noisy enough that some FIBs fail their CRC ---
    ens = synth.Ensemble(seed=0x601D, n_frames=4)
    iq = ens.iq()
    fic_soft, fic_fib, fic_ok = [], [], []
    for f, snr in zip(range(4), (14.0, 5.0, 3.0, 1.5)):
        rx = synth.channel(iq[f], snr_db=snr, cfo=0.11 / 2048, rng=rng)
        soft, _, _, _ = O.ofdm_demod_frame(rx[synth.NB_NULL:], -0.11 / 2048)
        fib, ok = O.fic_decode(soft)
        fic_soft.append(soft[:9216].copy()); fic_fib.append(fib); fic_ok.append(ok)
    np.savez_compressed(os.path.join(HERE, "fic_cases.npz"), soft=np.stack(fic_soft), fib=np.stack(fic_fib),
                        ok=np.stack(fic_ok), truth=ens.fibs)
    # --- Viterbi: per profile, 4 codewords: clean, noisy, pure noise, saturated noise ---
    out = {}
    for (opt, lvl, br) in PROFILES:
        mask, kept, nsteps, _ = O.eep_puncture_mask(opt, lvl, br)
        bits = rng.integers(0, 2, nsteps - 6, dtype=np.uint8)
        tx = np.where(O.conv_encode(bits)[mask.astype(bool)] > 0, 100.0, -100.0)
        cw = np.stack([
            np.clip(tx, -127, 127),
            np.clip(tx + rng.normal(0, 90, kept), -127, 127),
            rng.integers(-127, 128, kept).astype(np.float64),
            rng.choice([-127.0, 127.0, 0.0], kept),
        ]).astype(np.int8)
        want = np.stack([np.packbits(O.viterbi(O.depuncture(c, mask))) for c in cw])
        key = "eep%d_%d_%d" % (opt, lvl, br)
        out[key + "_punct"] = cw
        out[key + "_bytes"] = want
    np.savez_compressed(os.path.join(HERE, "viterbi_cases.npz"), **out)
    # --- whole-frame regression hashes (inputs are regenerated from seeds at test time) ---
    hashes = {}
    for seed, snr, cfo in ((1, None, 0.0), (2, 20.0, 0.37 / 2048), (3, 10.0, -1.3 / 2048)):
        e = synth.Ensemble(seed=seed, n_frames=1)
        r = np.random.default_rng(seed)
        rx = synth.channel(e.iq()[0], snr_db=snr, cfo=cfo, rng=r)
        soft, _, _, _ = O.ofdm_demod_frame(rx[synth.NB_NULL:], -cfo)
        fib, ok = O.fic_decode(soft)
        hashes[str(seed)] = {"snr": snr, "cfo": cfo, "iq_sha256": hashlib.sha256(rx.tobytes()).hexdigest(),
                             "soft_sha256": hashlib.sha256(soft.tobytes()).hexdigest(),
                             "fib_sha256": hashlib.sha256(fib.tobytes()).hexdigest(), "crc_ok": ok.tolist()}
    json.dump(hashes, open(os.path.join(HERE, "frame_hashes.json"), "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1:] == ["uep"]:
        make_uep()
    else:
        main()
        make_uep()
