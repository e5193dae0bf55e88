"""DAB+ audio super-frame checks: Fire code, RS(120,110), AU table and CRC.
CPU: oracle vs the independent numpy builder.  GPU: dabgpu_dabplus_superframes bit-exact vs the oracle, clean and
with byte errors, and end to end from IQ through the MSC Viterbi."""
import numpy as np
import pytest

from dabgpu import synth
from oracle import oracle as O

CASES = [(64, 1, 0), (32, 0, 1), (96, 1, 1), (128, 0, 0), (8, 0, 1), (192, 1, 0), (288, 0, 1), (256, 1, 0)]   # (12-bit AU start addresses: 110 s <= 4095)


def test_rs_and_firecode_agree_with_builder():
    rng = np.random.default_rng(1)
    for _ in range(5):
        d = rng.integers(0, 256, 110, dtype=np.uint8)
        assert (synth.rs_parity(d) == O.rs_encode(d)).all()
        assert synth.firecode16(d[:9]) == O.firecode(d[:9])
    d = rng.integers(0, 256, 110, dtype=np.uint8)
    cw = np.concatenate([d, O.rs_encode(d)])
    assert O.rs_decode(cw)[1] == 0
    for ne in range(1, 6):
        e = cw.copy()
        idx = rng.choice(120, ne, replace=False)
        e[idx] ^= rng.integers(1, 256, ne, dtype=np.uint8)
        c, r = O.rs_decode(e)
        assert r == ne and (c == cw).all()
    e = cw.copy()
    e[rng.choice(120, 7, replace=False)] ^= 0x33
    assert O.rs_decode(e)[1] == -1 or not (O.rs_decode(e)[0] == cw).all()     # beyond t=5: flagged or miscorrected


def test_rs_codewords_have_zero_syndromes_from_the_definition():
    """ETSI TS 102 563 clause 6: RS(120,110) shortened from RS(255,245) over GF(2^8) with P(x) = x^8+x^4+x^3+x^2+1 and
    generator prod_{i=0..9} (x + alpha^i).  Straight from that definition (bitwise field multiplication, no table shared
    with the oracle or the builder): every encoded word, read as a polynomial with its first byte as the highest power,
    vanishes at alpha^0 .. alpha^9; a word with one byte changed does not."""
    def gmul(a, b):
        r = 0
        while b:
            if b & 1:
                r ^= a
            a <<= 1
            if a & 0x100:
                a ^= 0x11D
            b >>= 1
        return r
    def evaluate(word, x):
        acc = 0
        for byte in word:                                        # Horner, highest power first
            acc = gmul(acc, x) ^ int(byte)
        return acc
    roots, a = [], 1
    for _ in range(10):
        roots.append(a)
        a = gmul(a, 2)
    rng = np.random.default_rng(5)
    for _ in range(4):
        d = rng.integers(0, 256, 110, dtype=np.uint8)
        cw = np.concatenate([d, O.rs_encode(d)])
        assert all(evaluate(cw, r) == 0 for r in roots)
        cw[int(rng.integers(0, 120))] ^= 0x5A
        assert any(evaluate(cw, r) != 0 for r in roots)


def test_firecode_is_the_remainder_by_its_generator():
    """TS 102 563 clause 5.2: the header's 16-bit Fire code word is computed over the 9 bytes that follow it with
    G(x) = (x^11 + 1)(x^5 + x^3 + x^2 + x + 1) = x^16+x^14+x^13+x^12+x^11+x^5+x^3+x^2+x+1, register started at zero:
    plain polynomial division over GF(2), written out bit by bit."""
    G = (1 << 16) | (1 << 14) | (1 << 13) | (1 << 12) | (1 << 11) | (1 << 5) | (1 << 3) | (1 << 2) | (1 << 1) | 1
    # (x^11 + 1)(x^5 + x^3 + x^2 + x + 1), multiplied out here rather than trusted
    a, b, prod = (1 << 11) | 1, 0b101111, 0
    for i in range(12):
        if (a >> i) & 1:
            prod ^= b << i
    assert prod == G
    rng = np.random.default_rng(6)
    for _ in range(8):
        d = rng.integers(0, 256, 9, dtype=np.uint8)
        rem = 0
        for byte in d:
            for bit in range(7, -1, -1):
                rem = (rem << 1) | ((int(byte) >> bit) & 1)
                if rem & (1 << 16):
                    rem ^= G
        for _ in range(16):                                      # the message times x^16
            rem <<= 1
            if rem & (1 << 16):
                rem ^= G
        assert O.firecode(d) == rem


@pytest.mark.parametrize("bitrate,dac_rate,sbr", CASES)
def test_oracle_superframe(bitrate, dac_rate, sbr):
    rng = np.random.default_rng(bitrate)
    sf, starts, aus = synth.build_superframe(rng, bitrate, dac_rate, sbr)
    s = bitrate // 8
    c, st, au = O.dabplus_superframe(sf, s)
    n = len(aus)
    assert st.tolist() == [1, 0, 0, n, (1 << n) - 1] and au[:n + 1].tolist() == starts
    # two byte errors in every column are repaired
    e = sf.copy()
    for j in range(s):
        e[j + s * rng.choice(120, 2, replace=False)] ^= 0xA5
    c, st, au = O.dabplus_superframe(e, s)
    assert (c == sf).all() and st.tolist() == [1, 2 * s, 0, n, (1 << n) - 1]
    # a corrupted header byte beyond repair of the Fire code (no RS help: damage 6 bytes of column 2's codeword)
    bad = sf.copy()
    bad[2::s][:6] ^= 0xFF
    _, st, _ = O.dabplus_superframe(bad, s)
    assert st[2] >= 1 or st[0] == 0
    # an all-zero window (silence, erasures) is NOT a super-frame although a zero header matches a zero check word
    _, st, _ = O.dabplus_superframe(np.zeros_like(sf), s)
    assert st[0] == 0 and st[3] == 0


def _noisy_batch(rng):
    sfs, truth = [], []
    for (br, dr, sb) in [(64, 1, 0)] * 7:
        sf, starts, aus = synth.build_superframe(rng, br, dr, sb)
        truth.append(sf.copy())
        sfs.append(sf)
    sfs = np.stack(sfs)
    s = 8
    sfs[1, rng.choice(960, 10, replace=False)] ^= 0x11          # few scattered errors
    for j in range(s):
        sfs[2, j + s * rng.choice(120, 5, replace=False)] ^= 0xC3   # 5 per column: the limit
    sfs[3, 0:960:8][:7] ^= 0x5A                                 # 7 errors in column 0: uncorrectable
    sfs[4, 20] ^= 0x01                                          # one bit inside an AU
    sfs[5] = rng.integers(0, 256, 960, dtype=np.uint8)          # garbage
    sfs[6] = 0                                                  # all zero: must not pass the Fire code
    return sfs, truth


@pytest.mark.gpu
def test_gpu_superframes_match_oracle(ctx):
    rng = np.random.default_rng(7)
    sfs, truth = _noisy_batch(rng)
    out, st = ctx.dabplus_superframes(sfs, 64)
    for i in range(sfs.shape[0]):
        c, ost, oau = O.dabplus_superframe(sfs[i], 8)
        assert (out[i] == c[:880]).all(), i
        got = [int(st["firecode_ok"][i]), int(st["rs_corrected"][i]), int(st["rs_uncorrectable"][i]),
               int(st["num_aus"][i]), int(st["au_crc_mask"][i])]
        assert got == ost.tolist(), (i, got, ost)
        assert st["au_start"][i].tolist() == oau.tolist()
    assert (out[2] == truth[2][:880]).all() and st["rs_corrected"][2] == 40
    assert st["rs_uncorrectable"][3] >= 1
    assert st["firecode_ok"][6] == 0 and st["num_aus"][6] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("bitrate,dac_rate,sbr", CASES)
def test_gpu_superframe_profiles(ctx, bitrate, dac_rate, sbr):
    rng = np.random.default_rng(bitrate + 1)
    sf, starts, aus = synth.build_superframe(rng, bitrate, dac_rate, sbr)
    s = bitrate // 8
    e = sf.copy()
    for j in range(s):
        e[j + s * rng.choice(120, 3, replace=False)] ^= 0x77
    out, st = ctx.dabplus_superframes(e[None, :], bitrate)
    n = len(aus)
    assert (out[0] == sf[:110 * s]).all()
    assert (st["firecode_ok"][0], st["rs_corrected"][0], st["rs_uncorrectable"][0], st["num_aus"][0],
            st["au_crc_mask"][0]) == (1, 3 * s, 0, n, (1 << n) - 1)
    assert st["au_start"][0][:n + 1].tolist() == starts


@pytest.mark.gpu
def test_gpu_access_unit_crcs_at_extreme_lengths(ctx):
    """The kernel checks an access unit's CRC on all 64 lanes (pieces of ceil(len / 64) bytes, padded at the front): units of 1, 2,
    64, 65 and 128 payload bytes and one long one, each clean and each with one byte wrong, against the oracle."""
    rng = np.random.default_rng(11)
    first = 11                                                    # six access units at 48 kHz without SBR
    cuts = [first + 3, first + 3 + 4, first + 7 + 66, first + 73 + 67, first + 140 + 130]
    sf, starts, aus = synth.build_superframe(rng, 64, 1, 0, cuts=cuts)
    assert [b - a - 2 for a, b in zip(starts, starts[1:])] == [1, 2, 64, 65, 128, 880 - starts[5] - 2]
    batch = [sf]
    for a in range(6):
        e = sf.copy()
        e[starts[a] + (starts[a + 1] - starts[a] - 2) // 2] ^= 0x40   # inside unit a's payload; the RS parity is made to agree
        for j in range(8):                                            # (the code would repair a lone error)
            e[110 * 8 + j::8] = synth.rs_parity(e[j:110 * 8:8])
        batch.append(e)
    out, st = ctx.dabplus_superframes(np.stack(batch), 64)
    for i in range(len(batch)):
        c, ost, oau = O.dabplus_superframe(batch[i], 8)
        got = [int(st["firecode_ok"][i]), int(st["rs_corrected"][i]), int(st["rs_uncorrectable"][i]), int(st["num_aus"][i]),
               int(st["au_crc_mask"][i])]
        assert got == ost.tolist() and (out[i] == c[:880]).all(), (i, got, ost)
        assert got[4] == (0x3F if i == 0 else 0x3F ^ (1 << (i - 1))), (i, got)


@pytest.mark.gpu
def test_end_to_end_iq_to_access_units(ctx):
    """IQ -> OFDM -> MSC Viterbi -> super-frame: the subchannel carries real DAB+ super-frames."""
    import dabgpu
    rng = np.random.default_rng(11)
    n_frames = 10                                   # 40 logical frames = 8 super-frames
    ens = synth.Ensemble(seed=5, n_frames=n_frames)
    sfs = [synth.build_superframe(rng, 64, 1, 0)[0] for _ in range(8)]
    payload = np.concatenate(sfs).reshape(40, 192)
    # re-encode the subchannel with this payload (cyclic interleaving as in Ensemble)
    coded = np.stack([synth.msc_encode_lf(payload[r], ens.mask) for r in range(40)])
    tx = synth.time_interleave(coded, cyclic=True)
    bits = ens.frame_bits.copy()
    for f in range(n_frames):
        for c in range(4):
            a = synth.NB_FIC_BITS + c * synth.NB_CIF_BITS
            bits[f, a:a + ens.size_cu * 64] = tx[4 * f + c]
    iq = np.stack([synth.modulate_frame(bits[f]) for f in range(n_frames)])
    rx = synth.channel(iq.ravel(), snr_db=11.0, rng=rng).reshape(n_frames, -1)
    soft, _, _ = ctx.ofdm_demod_frames(np.ascontiguousarray(rx[:, synth.NB_NULL:]))
    sc = dabgpu.subchannel(0, 64, level=3)
    lf, _ = ctx.msc_decode(sc, soft, n_streams=1)          # [1][40][192]; entry t = logical frame t-15
    got = lf[0, 15:40]                                      # logical frames 0..24 = super-frames 0..4
    out, st = ctx.dabplus_superframes(got.reshape(5, 960), 64)
    assert st["firecode_ok"].all() and (st["rs_uncorrectable"] == 0).all()
    assert (st["au_crc_mask"] == 63).all() and (st["num_aus"] == 6).all()
    for i in range(5):
        assert (out[i] == sfs[i][:880]).all()
    # a mis-aligned start (one logical frame late) is detected by the Fire code
    _, st_bad = ctx.dabplus_superframes(lf[0, 16:36].reshape(4, 960), 64)
    assert not st_bad["firecode_ok"].any()
