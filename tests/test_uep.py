"""UEP sub-channels (EN 300 401 clause 11.3.1, the 64 protection profiles of the FIG 0/1 short form).
The table is restated from memory in three places (product, oracle, transmitter); what pins it is that all 64 rows
satisfy the block-count and bit-budget identities exactly, and that the three copies agree."""
import ctypes as C

import numpy as np
import pytest

import dabgpu
from dabgpu import synth
from oracle import oracle as O

CASES = [0, 4, 13, 15, 33, 63]        # 32k/5; 32k/1 (pad 4); 56k/2 (pad 8); 64k/4 (three runs); 128k/5; 384k/1 (416 CU)


def test_table_identities_and_three_copies_agree(built):
    assert len(synth._UEP_TABLE) == 64
    seen = set()
    for i in range(64):
        blocks, cu, pad, br = synth.uep_profile(i)            # asserts both identities
        m, cu2 = synth.uep_mask(i)
        om, kept, nsteps, ocu = O.uep_puncture_mask(i)
        assert (m == om).all() and cu == cu2 == ocu and kept + pad == cu * 64 and nsteps == br * 24 + 6
        obr, olv, _ = O.uep_profile(i)
        sc = dabgpu.uep_subchannel(i, 0)                      # product copy, through the C ABI (no GPU needed)
        assert (sc.is_uep, sc.bitrate_kbps, sc.protection_level, sc.length) == (1, br, synth._UEP_TABLE[i][1], cu)
        assert (obr, olv) == (br, sc.protection_level)
        assert dabgpu.lib().dabgpu_subchannel_bytes(C.byref(sc)) == br * 3
        seen.add((br, sc.protection_level))
        # table-index order: by bit rate, then from the weakest (5) to the strongest (1) protection level
        if i:
            pb, pl = synth._UEP_TABLE[i - 1][:2]
            assert br > pb or (br == pb and sc.protection_level < pl)
    assert len(seen) == 64
    L = dabgpu.lib()
    sc = dabgpu.Subchannel()
    assert L.dabgpu_uep_subchannel(64, 0, C.byref(sc)) < 0 and L.dabgpu_uep_subchannel(-1, 0, C.byref(sc)) < 0
    assert L.dabgpu_uep_subchannel(63, 864 - 415, C.byref(sc)) < 0        # does not fit the CIF
    bad = dabgpu.Subchannel(0, 16, 1, 0, 5, 40)                           # 40 kbit/s is not a UEP rate
    assert L.dabgpu_subchannel_bytes(C.byref(bad)) < 0
    bad = dabgpu.Subchannel(0, 17, 1, 0, 5, 32)                           # wrong size for 32 kbit/s level 5
    assert L.dabgpu_subchannel_bytes(C.byref(bad)) < 0


@pytest.mark.parametrize("index", [0, 4, 15])
def test_oracle_decodes_what_the_transmitter_sent(index):
    e = synth.Ensemble(seed=900 + index, n_frames=5, uep_index=index, start_cu=7)
    mask, kept, nsteps, cu = O.uep_puncture_mask(index)
    nbits = cu * 64
    rng = np.random.default_rng(index)
    bits = e.frame_bits[:, synth.NB_FIC_BITS:].reshape(20, synth.NB_CIF_BITS)[:, 7 * 64:7 * 64 + nbits]
    soft = np.where(bits > 0, 127, -127).astype(np.int16) + rng.integers(-90, 91, bits.shape)
    soft = np.clip(soft, -127, 127).astype(np.int8)
    for t in range(15, 20):
        lf = O.msc_decode_lf(O.time_deinterleave(soft[t - 15:t + 1])[:kept], mask, nsteps)
        assert (lf == e.msc_bytes[t - 15]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("index", CASES)
def test_gpu_uep_subchannel_bit_exact(ctx, index):
    start = 3 if index != 63 else 448
    n_frames = 5
    e = synth.Ensemble(seed=500 + index, n_frames=n_frames, uep_index=index, start_cu=start)
    rng = np.random.default_rng(index)
    rx = synth.channel(e.iq().ravel(), snr_db=8.0, rng=rng).reshape(n_frames, -1)
    soft = ctx.ofdm_demod_frames(np.ascontiguousarray(rx[:, synth.NB_NULL:]))[0]
    sc = dabgpu.uep_subchannel(index, start)
    nbits = sc.length * 64
    out, hist = ctx.msc_decode(sc, soft, n_streams=1, want_history=True)
    mask, kept, nsteps, _ = O.uep_puncture_mask(index)
    cifs = soft[:, synth.NB_FIC_BITS:].reshape(4 * n_frames, synth.NB_CIF_BITS)[:, start * 64:start * 64 + nbits]
    padded = np.concatenate([np.zeros((15, nbits), np.int8), cifs])
    for t in range(4 * n_frames):
        want = O.msc_decode_lf(O.time_deinterleave(padded[t:t + 16])[:kept], mask, nsteps)
        assert (out[0, t] == want).all(), t
        if t >= 15 and index != 63:                         # 384 kbit/s level 1 at 8 dB: parity yes, error-free no
            assert (want == e.msc_bytes[t - 15]).all()
    assert (hist[0] == cifs[-15:]).all()


@pytest.mark.gpu
def test_gpu_uep_golden_vectors(ctx):
    from conftest import golden_path
    from golden.make_golden import UEP_INDICES
    d = np.load(golden_path("uep_cases.npz"))
    for idx in UEP_INDICES:
        mask = O.uep_puncture_mask(idx)[0]
        assert (ctx.viterbi(d["uep%d_punct" % idx], mask) == d["uep%d_bytes" % idx]).all()
