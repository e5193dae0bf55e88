"""Known-answer tests of the CPU oracle against the independent numpy transmitter: the
reference has no tests or vectors, so truth = transmitted bits."""
import numpy as np
import pytest

from dabgpu import synth
from oracle import oracle as O


def test_fft_matches_numpy():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(2048) + 1j * rng.standard_normal(2048)).astype(np.complex64)
    ref = np.fft.fft(x.astype(np.complex128))
    got = O.fft2048(x)
    assert np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max() * 11     # ~log2(N) ulp growth
    e = np.zeros(2048, np.complex64)
    e[1] = 1
    assert np.allclose(O.fft2048(e), np.exp(-2j * np.pi * np.arange(2048) / 2048), atol=1e-6)


@pytest.mark.parametrize("snr,cfo", [(None, 0.0), (20.0, 0.37 / 2048), (9.0, -0.21 / 2048), (15.0, 3.25 / 2048)])
def test_frame_roundtrip(ensemble, ensemble_iq, snr, cfo):
    rng = np.random.default_rng(5)
    rx = synth.channel(ensemble_iq.ravel(), snr_db=snr, cfo=cfo, rng=rng).reshape(ensemble_iq.shape)
    for f in range(2):
        soft, _, cyc, dq = O.ofdm_demod_frame(rx[f, synth.NB_NULL:], -cfo, want_cyc=True, want_dqpsk=True)
        fib, ok = O.fic_decode(soft)
        assert ok.all() and (fib == ensemble.fibs[f]).all()
        if snr is None:
            assert ((soft > 0).astype(np.uint8) == ensemble.frame_bits[f]).all()
            assert np.abs(soft).max() == 127
            # residual cyclic phase ~ 0 once the offset is corrected
            assert np.abs(np.angle(cyc)).max() < 1e-3
        assert np.abs(soft.astype(int)).max() <= 127


def test_cyclic_prefix_correlation_measures_fine_offset(ensemble_iq):
    cfo = 0.2 / 2048          # 0.2 carrier spacings, uncorrected
    rx = synth.channel(ensemble_iq[0], cfo=cfo)
    _, _, cyc, _ = O.ofdm_demod_frame(rx[synth.NB_NULL:], 0.0, want_cyc=True)
    est = np.angle(cyc).mean() / (2 * np.pi * 2048)
    assert abs(est - cfo) < 0.01 / 2048


def test_uncorrected_integer_offset_breaks_and_correction_fixes(ensemble, ensemble_iq):
    cfo = 2.0 / 2048
    rx = synth.channel(ensemble_iq[0], cfo=cfo)
    soft_bad, _, _, _ = O.ofdm_demod_frame(rx[synth.NB_NULL:], 0.0)
    assert not O.fic_decode(soft_bad)[1].all()
    soft, _, _, _ = O.ofdm_demod_frame(rx[synth.NB_NULL:], -cfo)
    assert O.fic_decode(soft)[1].all()


def test_msc_roundtrip_through_time_deinterleaver(ensemble, ensemble_iq):
    rng = np.random.default_rng(9)
    rx = synth.channel(ensemble_iq.ravel(), snr_db=12.0, rng=rng).reshape(ensemble_iq.shape)
    softs = np.stack([O.ofdm_demod_frame(rx[f, synth.NB_NULL:])[0] for f in range(5)])
    cifs = softs[:, synth.NB_FIC_BITS:].reshape(20, synth.NB_CIF_BITS)[:, :ensemble.size_cu * 64]
    for t in range(15, 20):
        de = O.time_deinterleave(cifs[t - 15:t + 1])
        assert (O.msc_decode_lf(de, ensemble.mask, 64 * 24 + 6) == ensemble.msc_bytes[t - 15]).all()


@pytest.mark.parametrize("option,level,bitrate", [(0, 1, 8), (0, 2, 8), (0, 2, 16), (0, 4, 8), (0, 3, 128), (1, 1, 32), (1, 4, 64)])
def test_viterbi_decodes_every_profile_family(option, level, bitrate):
    rng = np.random.default_rng(bitrate + level)
    mask, kept, nsteps, _ = O.eep_puncture_mask(option, level, bitrate)
    bits = rng.integers(0, 2, nsteps - 6, dtype=np.uint8)
    tx = O.conv_encode(bits)[mask.astype(bool)]
    soft = np.where(tx > 0, 127, -127).astype(np.int8)
    assert (O.viterbi(O.depuncture(soft, mask)) == bits).all()
    # a few well-separated erasures and flips must be corrected at these rates
    soft2 = soft.copy()
    idx = np.arange(20, kept - 20, max(kept // 8, 97))
    soft2[idx] = -soft2[idx]
    if level <= 3:
        assert (O.viterbi(O.depuncture(soft2, mask)) == bits).all()


def test_viterbi_degenerate_inputs_are_deterministic():
    z = np.zeros(4 * 30, np.int8)
    assert not O.viterbi(z).any()                         # all ties -> older-bit-0 survivor -> zeros
    rng = np.random.default_rng(2)
    g = rng.integers(-127, 128, 4 * 774, dtype=np.int8)
    assert (O.viterbi(g) == O.viterbi(g.copy())).all()


def test_time_interleaver_is_inverse_of_deinterleaver():
    rng = np.random.default_rng(1)
    coded = rng.integers(-127, 128, size=(32, 128)).astype(np.int8)
    tx = synth.time_interleave(coded, cyclic=True)
    for r in range(32):
        rows = np.stack([tx[(r + k) % 32] for k in range(16)])
        assert (O.time_deinterleave(rows) == coded[r]).all()


def test_decision_directed_frequency_error_tracks_the_cyclic_prefix_estimate(built):
    """oracle.dd_error (fourth power of the differential symbols over the 256 centre carriers of a symbol, branch picked
    by the PRS's cyclic-prefix correlation) against the all-symbol cyclic-prefix estimate and the injected residual,
    through noise and echoes, over the whole +-0.5 carrier range."""
    import numpy as np
    from dabgpu import synth
    from oracle import oracle as O
    e = synth.Ensemble(seed=21, n_frames=4)
    for resid, snr, paths in ((0.05, 12.0, None), (-0.08, 8.0, None), (0.03, 15.0, [(0, 1.0), (120, 0.8j)]), (0.0, 10.0, None),
                              (0.31, 12.0, None), (-0.22, 10.0, [(0, 1.0), (60, 0.5)]), (0.45, 15.0, None), (-0.1, 9.0, None)):
        rng = np.random.default_rng(int(1000 * abs(resid)) + 7)
        rx = synth.channel(e.iq().ravel(), snr_db=snr, cfo=resid / 2048, rng=rng, paths=paths).reshape(4, -1)
        fr = np.ascontiguousarray(rx[0, synth.NB_NULL - 8:synth.NB_NULL - 8 + 76 * 2552])
        _, dd = O.ofdm_demod_frame_dd(fr, 0.0)
        _, _, cyc, _ = O.ofdm_demod_frame(fr, 0.0, want_cyc=True)
        e_dd = float(O.dd_error(dd)) * 2048
        e_cp = float(np.angle(cyc.astype(np.complex128)).mean() / (2 * np.pi * 2048)) * 2048
        # (approaching half a carrier the carriers leak into each other: the decisions, and with them the fourth-power
        # estimate, degrade -- a loop still lands inside the accurate region with its first step)
        assert abs(e_dd - resid) < (0.004 if abs(resid) <= 0.35 else 0.05) and abs(e_cp - resid) < 0.006, (resid, e_dd, e_cp)
    # a sample clock 150 ppm off rotates carrier k by 2 pi k 1.5e-4 2552/2048 per symbol: the outer carriers' fourth
    # powers would point the other way, the 256 centre carriers the estimator uses do not care
    for ppm in (150.0, -150.0):
        rx = synth.channel(e.iq().ravel(), snr_db=15.0, cfo=0.04 / 2048, rng=np.random.default_rng(3), sco_ppm=ppm).reshape(-1)
        start = int(round((synth.NB_NULL + 12) / (1 + ppm * 1e-6))) - 8          # resample()'s group delay is 12 samples
        fr = np.ascontiguousarray(rx[start:start + 76 * 2552])
        _, dd = O.ofdm_demod_frame_dd(fr, 0.0)
        assert abs(float(O.dd_error(dd)) * 2048 - 0.04) < 0.004, (ppm, float(O.dd_error(dd)) * 2048)
    # the level of the input does not matter (every term is scaled as the quantiser scales it): 16-bit sample values
    # would overflow a plain fourth power of X conj X in single precision
    _, dd_big = O.ofdm_demod_frame_dd(fr * np.float32(30000.0), 0.0)
    assert np.isfinite(dd_big).all() and abs(float(O.dd_error(dd_big)) - float(O.dd_error(dd))) * 2048 < 1e-5
