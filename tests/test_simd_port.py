"""oracle/simd_port.c -- the second CPU implementation bench.py times beside the oracle (`cpu_baseline.simd_port`):
table-driven NCO, four-step FFT in AVX loops, 16-bit saturating AVX2 Viterbi, built -O3 -march=native -ffast-math for
the host it runs on.  It is NOT the checker: its decoder is not bit-identical to the oracle's on noise.  What it must
do is decode to the TRANSMITTED data, and demodulate to the oracle's soft bits within one LSB."""
import numpy as np
import pytest

from dabgpu import synth
from oracle import oracle as O


@pytest.fixture(scope="module")
def rx():
    e = synth.Ensemble(seed=0x51D, n_frames=4)
    rng = np.random.default_rng(0x51D)
    cfo = -0.37 / 2048
    x = synth.channel(e.iq().ravel(), snr_db=11.0, cfo=cfo, rng=rng, paths=[(0, 1.0), (50, 0.4j)]).reshape(4, -1)
    return e, np.ascontiguousarray(x[:, synth.NB_NULL - 8:synth.NB_NULL - 8 + 76 * 2552]), -cfo


def test_simd_port_demodulates_like_the_oracle_and_decodes_the_truth(built, rx):
    e, frames, fo = rx
    softs = []
    for f in range(4):
        soft = O.simd_ofdm_demod_frame(frames[f], fo)
        osoft = O.ofdm_demod_frame(frames[f], fo)[0]
        d = np.abs(soft.astype(np.int32) - osoft.astype(np.int32))
        assert d.max() <= 1 and (d != 0).mean() < 0.1            # fast-math FFT vs the oracle's: last-ulp differences only
        fib, ok = O.simd_fic_decode(soft)
        assert ok.all() and (fib == e.fibs[f]).all()
        softs.append(soft)
    cifs = np.stack(softs)[:, 9216:].reshape(16, 55296)[:, :e.size_cu * 64]
    lf = O.simd_msc_decode_lf(O.time_deinterleave(cifs), e.mask, 64 * 24 + 6)
    assert (lf == e.msc_bytes[0]).all()
    assert "avx" in O.simd_isa() or "scalar" in O.simd_isa()


def test_simd_viterbi_handles_erasures_and_heavy_puncturing(built):
    """UEP / EEP profiles with long punctured runs (erasures cost the same on every branch) and a clean codeword."""
    rng = np.random.default_rng(3)
    for option, level, bitrate in ((0, 4, 24), (1, 4, 64), (0, 1, 8)):
        mask, kept, nsteps, _ = O.eep_puncture_mask(option, level, bitrate)
        bits = rng.integers(0, 2, nsteps - 6, dtype=np.uint8)
        tx = np.where(O.conv_encode(bits)[mask.astype(bool)] > 0, 90.0, -90.0)
        punct = np.clip(tx + rng.normal(0, 45 if level == 4 else 70, kept), -127, 127).astype(np.int8)
        punct[punct == 0] = 1
        out = O.simd_msc_decode_lf(punct, mask, nsteps)
        prbs = np.packbits(O.prbs(nsteps - 6))
        assert (out ^ prbs == np.packbits(bits)).all(), (option, level, bitrate)


def test_simd_bench_harness_runs(built, rx):
    e, frames, fo = rx
    fo_arr = np.full(4, fo, np.float32)
    k, t = O.bench_frames_timed(frames, fo_arr, 0.3, 2, e.mask, 64 * 24 + 6, e.size_cu * 64, simd=True)
    assert k >= 2 and 0.2 < t < 5.0
    k, t = O.bench_ofdm_only_timed(frames, fo_arr, 0.2, simd=True)
    assert k >= 1
    k, t = O.bench_pipeline_timed(frames, fo_arr, 0.3, e.mask, 64 * 24 + 6, e.size_cu * 64, simd=True)
    assert k >= 1
    ko, _ = O.bench_frames_timed(frames, fo_arr, 0.3, 1, e.mask, 64 * 24 + 6, e.size_cu * 64)
    assert ko >= 1
