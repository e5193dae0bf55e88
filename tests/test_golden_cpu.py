"""The oracle against the committed golden vectors (tests/golden, made by make_golden.py)."""
import hashlib
import json

import numpy as np

from conftest import golden_path
from dabgpu import synth
from oracle import oracle as O
from golden.make_golden import PROFILES, UEP_INDICES


def test_fic_cases():
    d = np.load(golden_path("fic_cases.npz"))
    for f in range(d["soft"].shape[0]):
        fib, ok = O.fic_decode(d["soft"][f])
        assert (fib == d["fib"][f]).all() and (ok == d["ok"][f]).all()
        # wherever the CRC passes the FIB equals what was transmitted
        assert (fib[ok.astype(bool)] == d["truth"][f][ok.astype(bool)]).all()
    assert not d["ok"][3].all() and d["ok"][0].all()       # the set holds failing CRCs too


def test_viterbi_cases():
    d = np.load(golden_path("viterbi_cases.npz"))
    for (opt, lvl, br) in PROFILES:
        mask = O.eep_puncture_mask(opt, lvl, br)[0]
        key = "eep%d_%d_%d" % (opt, lvl, br)
        for c, want in zip(d[key + "_punct"], d[key + "_bytes"]):
            assert (np.packbits(O.viterbi(O.depuncture(c, mask))) == want).all()


def test_uep_cases():
    d = np.load(golden_path("uep_cases.npz"))
    for idx in UEP_INDICES:
        mask = O.uep_puncture_mask(idx)[0]
        for c, want in zip(d["uep%d_punct" % idx], d["uep%d_bytes" % idx]):
            assert (np.packbits(O.viterbi(O.depuncture(c, mask))) == want).all()
        assert (d["uep%d_bytes" % idx][0] == d["uep%d_truth" % idx]).all()      # the clean codeword decodes to what was sent


def test_frame_hashes():
    h = json.load(open(golden_path("frame_hashes.json")))
    for seed, rec in h.items():
        e = synth.Ensemble(seed=int(seed), n_frames=1)
        r = np.random.default_rng(int(seed))
        rx = synth.channel(e.iq()[0], snr_db=rec["snr"], cfo=rec["cfo"], rng=r)
        if hashlib.sha256(rx.tobytes()).hexdigest() != rec["iq_sha256"]:
            continue      # numpy/libm produced different float IQ on this host: input not reproducible
        soft, _, _, _ = O.ofdm_demod_frame(rx[synth.NB_NULL:], -rec["cfo"])
        fib, ok = O.fic_decode(soft)
        assert hashlib.sha256(fib.tobytes()).hexdigest() == rec["fib_sha256"]
        assert ok.tolist() == rec["crc_ok"]
