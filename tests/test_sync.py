"""Frame synchronisation on the phase reference symbol: coarse frequency (integer carriers) and
fine time.  CPU: the oracle recovers known offsets.  GPU: dabgpu_sync_prs returns the same integers as the oracle
(and the truth), ratios within tolerance."""
import numpy as np
import pytest

from dabgpu import synth
from oracle import oracle as O

CASES = [(0, 0, None), (5, 100, 15.0), (-37, 17, 10.0), (120, 300, 8.0), (0, -40, 20.0), (-199, 250, 12.0)]


def _candidates(seed=3):
    e = synth.Ensemble(seed, n_frames=1)
    iq = e.iq()[0]
    rng = np.random.default_rng(seed)
    rows, truth = [], []
    for cfo_c, early, snr in CASES:
        # a fractional part as well: the fine loop has not converged when coarse sync runs
        rx = synth.channel(iq, snr_db=snr, cfo=(cfo_c + 0.2) / 2048, rng=rng)
        st = synth.NB_NULL - early
        rows.append(rx[st:st + synth.NB_SYM])
        truth.append((cfo_c, early))
    return np.stack(rows), truth


def test_oracle_recovers_known_offsets():
    rows, truth = _candidates()
    for r, (k, t) in zip(rows, truth):
        kk, tt, ptm, cptm = O.sync_prs(r, 0.0, 200)
        assert (kk, tt) == (k, t)
        assert ptm > 100 and cptm > 10             # > 20 dB impulse peak, clear coarse peak


def test_oracle_noise_only_has_no_peak():
    rng = np.random.default_rng(0)
    n = (rng.standard_normal(2552) + 1j * rng.standard_normal(2552)).astype(np.complex64)
    _, _, ptm, _ = O.sync_prs(n, 0.0, 100)
    assert ptm < 30                                 # nothing near the 20 dB (x100) threshold


def test_fine_frequency_input_is_applied():
    rows, truth = _candidates()
    # derotating by the true coarse offset moves the estimate to zero
    k, t = truth[1]
    kk, tt, _, _ = O.sync_prs(rows[1], -k / 2048, 200)
    assert (kk, tt) == (0, t)


@pytest.mark.gpu
def test_gpu_sync_matches_oracle(ctx):
    rows, truth = _candidates()
    got = ctx.sync_prs(rows, None, 200)
    for i, (r, (k, t)) in enumerate(zip(rows, truth)):
        kk, tt, ptm, cptm = O.sync_prs(r, 0.0, 200)
        assert (int(got["coarse_carriers"][i]), int(got["time_offset"][i])) == (kk, tt) == (k, t)
        assert abs(got["peak_to_mean"][i] - ptm) <= 1e-3 * ptm
        assert abs(got["coarse_peak_to_mean"][i] - cptm) <= 1e-3 * cptm
    fo = np.array([-k / 2048 for k, _ in truth], np.float32)
    got2 = ctx.sync_prs(rows, fo, 64)
    assert (got2["coarse_carriers"] == 0).all()
    assert got2["time_offset"].tolist() == [t for _, t in truth]


@pytest.mark.gpu
def test_gpu_sync_batch_and_arguments(ctx):
    import dabgpu
    rows, truth = _candidates()
    big = np.tile(rows, (50, 1))                     # 300 candidates in one launch
    got = ctx.sync_prs(big, None, 200)
    assert got["coarse_carriers"].reshape(50, -1).tolist() == [[k for k, _ in truth]] * 50
    assert got["time_offset"].reshape(50, -1).tolist() == [[t for _, t in truth]] * 50
    with pytest.raises(dabgpu.DabGpuError):
        ctx.sync_prs(rows, None, 5000)
    assert ctx.sync_prs(np.zeros((0, 2552), np.complex64)).size == 0
