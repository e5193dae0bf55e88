"""Acquisition on unaligned captures: null-symbol power-dip search, fractional frequency from the
cyclic prefix, coarse frequency + fine time on the PRS, then demodulation of the frames found where they lie.
CPU part: the oracle against known answers.  GPU part: the kernels against the oracle and against the
transmitted FIBs."""
import numpy as np
import pytest

from dabgpu import synth
from oracle import oracle as O

NULL = synth.NB_NULL
FRAME = synth.NB_FRAME_SAMPLES
SYMS = 76 * 2552


def capture(seed, n_frames, cut, tail, snr, cfo_carriers, gain=1.0):
    """A capture that starts `cut` samples into a frame and ends `tail` samples into another one.
    -> (iq, true first-PRS-sample positions of the whole frames, ensemble, index of the first whole frame)."""
    e = synth.Ensemble(seed=seed, n_frames=n_frames)
    iq = e.iq().ravel()
    x = np.concatenate([iq[cut:], iq[:tail]]) if tail else iq[cut:]
    rng = np.random.default_rng(seed)
    x = synth.channel(x, snr_db=snr, cfo=cfo_carriers / 2048.0, rng=rng) * np.float32(gain)
    first = 0 if cut <= 0 else 1
    starts = [k * FRAME + NULL - cut for k in range(first, n_frames) if k * FRAME + NULL - cut + SYMS + 512 <= x.size]
    if tail >= NULL + SYMS + 512:
        starts.append(n_frames * FRAME + NULL - cut)
    return x.astype(np.complex64), np.array(starts, np.int64), e, first


def test_oracle_block_norms_match_numpy():
    x, _, _, _ = capture(3, 1, 5000, 0, 20.0, 0.0)
    l1 = O.null_block_l1(x)
    ref = (np.abs(x.real.astype(np.float64)) + np.abs(x.imag.astype(np.float64)))[:l1.size * 64].reshape(-1, 64).sum(1)
    assert np.allclose(l1, ref, rtol=1e-5)


@pytest.mark.parametrize("snr,cfo,gain", [(25.0, 0.0, 1.0), (12.0, 3.3, 0.02), (8.0, -17.45, 40.0)])
def test_oracle_finds_every_whole_frame(snr, cfo, gain):
    x, starts, _, _ = capture(11, 4, 123457, 60001, snr, cfo, gain)
    cands = O.null_search(x)
    assert len(cands) == len(starts)
    assert (np.abs(cands - starts) <= 96).all()               # block resolution + noise
    for c, s in zip(cands, starts):
        r = O.acquire_candidate(x, c, margin=0)
        assert r.flags == 3
        assert abs(r.start - s) <= 1                          # sample-accurate timing
        assert r.coarse_carriers == int(np.round(cfo))        # signal sits cfo carriers high
        frac = cfo - np.round(cfo)
        assert abs(r.fine_offset * 2048 + frac) < 0.02        # correction = -fractional part
        assert abs(r.freq_offset * 2048 + cfo) < 0.02


def test_oracle_ignores_short_dips_and_silence():
    x, starts, _, _ = capture(5, 2, 0, 0, 20.0, 0.0)
    y = x.copy()
    y[50000:50000 + 640] = 0                                  # a 10-block dropout is not a null symbol
    assert len(O.null_search(y)) == len(O.null_search(x))
    y = x.copy()
    y[300000:] *= 1e-3                                        # the signal vanishes: no frame may be invented
    got = O.null_search(y)
    assert (got < 300000).all()
    assert len(O.null_search(np.zeros(4096, np.complex64))) == 0


def faded_capture(seed=17, n_frames=10, depth_db=14.0, hz=3.0):
    """A capture whose level swings by `depth_db` at `hz` (a deep slow fade), cut at an odd place."""
    e = synth.Ensemble(seed=seed, n_frames=4)
    tx = np.tile(e.iq().ravel(), (n_frames + 3) // 4)[:n_frames * FRAME]
    t = np.arange(tx.size) / 2.048e6
    lo = 10 ** (-depth_db / 20.0)
    gain = lo + (1.0 - lo) * 0.5 * (1.0 + np.cos(2 * np.pi * hz * t))
    rng = np.random.default_rng(seed)
    x = synth.channel(tx, snr_db=28.0, cfo=0.8 / 2048, rng=rng, gain=gain)[70001:]
    starts = np.array([k * FRAME + NULL - 70001 for k in range(1, n_frames) if k * FRAME + NULL - 70001 + SYMS + 512 <= x.size])
    return x, starts, e


def test_oracle_null_search_follows_the_local_level():
    """Thresholds relative to the capture's mean lose frames in a deep slow fade (the faded signal sits below
    thr_start x mean: one endless 'dip'); relative to the local level (level_chunk 256 = the default) every frame is
    found -- what the reference's running level average does for a stream."""
    x, starts, _ = faded_capture()
    local = O.null_search(x, level_chunk=256)
    whole = O.null_search(x, level_chunk=0)
    assert len(local) == len(starts) and (np.abs(local - starts) <= 96).all()
    assert len(whole) < len(starts)


# ------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def gctx(built):
    from conftest import make_ctx
    c = make_ctx(None, 16)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("snr,cfo,cut", [(25.0, 0.0, 100000), (12.0, 3.3, 77777), (9.0, -17.45, 150001)])
def test_gpu_acquire_matches_oracle(gctx, snr, cfo, cut):
    import dabgpu
    caps = [capture(21 + i, 3, cut + 999 * i, 50000, snr, cfo + 0.11 * i) for i in range(3)]
    n = min(c[0].size for c in caps)
    iq = np.stack([c[0][:n] for c in caps])
    cfg = dabgpu.acquire_cfg(timing_margin=0)
    frames, counts = gctx.acquire(iq, 6, cfg)
    for s in range(3):
        cands = O.null_search(iq[s], max_out=6)
        assert counts[s] == len(cands) >= 2
        for j, c in enumerate(cands):
            r = O.acquire_candidate(iq[s], c, margin=0)
            g = frames[s, j]
            assert (g["start"], g["coarse_carriers"], g["flags"]) == (r.start, r.coarse_carriers, r.flags)
            assert abs(g["fine_offset"] - r.fine_offset) <= 1e-9 + 1e-5 * abs(r.fine_offset)
            assert abs(g["peak_to_mean"] - r.peak_to_mean) <= 2e-3 * r.peak_to_mean
            assert abs(g["coarse_peak_to_mean"] - r.coarse_peak_to_mean) <= 2e-3 * r.coarse_peak_to_mean
        assert (frames[s, counts[s]:]["flags"] == 0).all() and (frames[s, counts[s]:]["start"] == -1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", [256, 64, 4096, 0])
def test_gpu_null_search_with_local_level_matches_oracle(gctx, chunk):
    """The local-level thresholds (dabgpu_acquire_cfg.level_chunk_blocks) on a deeply faded capture and on a plain one:
    the same candidates as the oracle's sequential search, for several chunk sizes and for the capture-mean rule."""
    import dabgpu
    cfg = dabgpu.acquire_cfg(timing_margin=0, level_chunk_blocks=chunk)
    xf, starts, _ = faded_capture()
    xp, _, _, _ = capture(33, 10, 70001, 0, 18.0, -1.7)
    n = min(xf.size, xp.size)
    iq = np.stack([xf[:n], xp[:n]])
    frames, counts = gctx.acquire(iq, 12, cfg)
    for s in range(2):
        cands = O.null_search(iq[s], max_out=12, level_chunk=chunk)
        assert counts[s] == len(cands), (chunk, s, counts[s], len(cands))
        for j, c in enumerate(cands):
            r = O.acquire_candidate(iq[s], c, margin=0)
            g = frames[s, j]
            assert (g["start"], g["coarse_carriers"], g["flags"]) == (r.start, r.coarse_carriers, r.flags), (chunk, s, j)
    if chunk in (64, 256):
        st = starts[starts + SYMS + 512 <= n]
        assert counts[0] == len(st) and (np.abs(frames[0, :counts[0]]["start"] - st) <= 2).all()
    with pytest.raises(dabgpu.DabGpuError):
        gctx.acquire(iq, 12, dabgpu.acquire_cfg(level_chunk_blocks=100))


@pytest.mark.gpu
@pytest.mark.parametrize("cut", [132072, 131072, 133728, 131500])
def test_gpu_null_search_across_segment_boundaries(gctx, cut):
    """The dip search scans segments of 2^20 samples in parallel and stitches them.  Captures of 12 frames (three
    segments) cut so that a null symbol straddles the first boundary, starts exactly on it, ends exactly on it, or has
    its first quiet block just before it: the frames found are the sequential search's (the oracle's plain loop)."""
    import dabgpu
    x, starts, _, _ = capture(31, 12, cut, 0, 20.0, 1.2)
    cfg = dabgpu.acquire_cfg(timing_margin=0)
    frames, counts = gctx.acquire(x[None, :], 16, cfg)
    cands = O.null_search(x, max_out=16)
    assert counts[0] == len(cands) == len(starts)
    for j, c in enumerate(cands):
        r = O.acquire_candidate(x, c, margin=0)
        g = frames[0, j]
        assert (g["start"], g["coarse_carriers"], g["flags"]) == (r.start, r.coarse_carriers, r.flags), j
        assert abs(int(g["start"]) - int(starts[j])) <= 8 and g["flags"] == 3


@pytest.mark.gpu
def test_gpu_unaligned_capture_to_fibs(gctx):
    """Whole chain with nothing known in advance: capture -> acquire -> demodulate in place -> FIC."""
    import torch
    import dabgpu
    n_streams, max_frames = 4, 5
    caps = [capture(40 + i, 4, 30001 + 17001 * i, 60000, 14.0, (-1) ** i * (2.0 + 0.37 * i), gain=0.5 + i)
            for i in range(n_streams)]
    n = min(c[0].size for c in caps)
    dev = torch.device("cuda", 0)
    iq = torch.from_numpy(np.stack([c[0][:n] for c in caps])).to(dev)
    frames = torch.zeros((n_streams, max_frames, 32), dtype=torch.uint8, device=dev)
    counts = torch.zeros(n_streams, dtype=torch.int32, device=dev)
    soft = torch.full((n_streams * max_frames, dabgpu.NB_FRAME_BITS), 7, dtype=torch.int8, device=dev)
    fib = torch.zeros((n_streams * max_frames, 12, 32), dtype=torch.uint8, device=dev)
    ok = torch.zeros((n_streams * max_frames, 12), dtype=torch.uint8, device=dev)
    gctx.acquire_dev(iq.data_ptr(), n, n_streams, n, max_frames, frames.data_ptr(), counts.data_ptr())
    gctx.ofdm_demod_acquired_dev(iq.data_ptr(), n, n_streams, max_frames, frames.data_ptr(), soft.data_ptr())
    gctx.fic_decode_dev(soft.data_ptr(), dabgpu.NB_FRAME_BITS, n_streams * max_frames, fib.data_ptr(), ok.data_ptr())
    gctx.sync()
    fr = frames.cpu().numpy().view(dabgpu.ACQUIRED_FRAME_DTYPE).reshape(n_streams, max_frames)
    cnt, soft_h = counts.cpu().numpy(), soft.cpu().numpy().reshape(n_streams, max_frames, -1)
    fib_h, ok_h = fib.cpu().numpy().reshape(n_streams, max_frames, 12, 32), ok.cpu().numpy().reshape(n_streams, max_frames, 12)
    odd = 0
    for s, (x, starts, e, first) in enumerate(caps):
        starts = starts[starts + SYMS + 512 <= n]
        assert cnt[s] == len(starts) == 3, (cnt[s], starts)
        for j in range(cnt[s]):
            assert fr[s, j]["flags"] == 3
            assert abs(fr[s, j]["start"] + 64 - starts[j]) <= 1          # default timing margin: 64 samples early
            odd += int(fr[s, j]["start"] & 1)
            assert ok_h[s, j].all()
            assert (fib_h[s, j] == e.fibs[first + j]).all()              # known answer: the transmitted FIBs
            # parity with the oracle demodulating the same samples with the same correction
            st = int(fr[s, j]["start"])
            osoft, _, _, _ = O.ofdm_demod_frame(x[st:st + SYMS], float(fr[s, j]["freq_offset"]))
            assert np.abs(soft_h[s, j].astype(np.int32) - osoft.astype(np.int32)).max() <= 1
        assert not soft_h[s, cnt[s]:].any()                              # unused entries: erased, not stale
    assert odd > 0                                                       # the 8-byte-aligned load path was exercised


def multipath_capture(seed, paths, snr, cfo_carriers, cut, ppm=0.0):
    e = synth.Ensemble(seed=seed, n_frames=4)
    rng = np.random.default_rng(seed)
    x = synth.channel(e.iq().ravel(), snr_db=snr, cfo=cfo_carriers / 2048.0, rng=rng, paths=paths, sco_ppm=ppm)[cut:]
    half = 12 if ppm else 0
    starts = np.array([(k * FRAME + NULL) * (1 + ppm * 1e-6) - half - cut for k in range(1, 4)])
    return x, starts[starts + SYMS + 512 <= x.size], e


@pytest.mark.gpu
@pytest.mark.parametrize("echo_db,delay", [(3.0, 200), (2.0, 330), (-4.0, 120), (0.0, 60)])
def test_gpu_acquisition_aligns_to_the_first_significant_path(gctx, echo_db, delay):
    """An echo up to 3 dB STRONGER than the first arrival, anywhere inside the cyclic prefix: acquisition (default
    cfg: first_path_rel 0.25) aligns the frame to the first path -- so the echo falls inside the prefix and the FIBs
    decode --, exactly as the oracle does on the same samples; with the reference's rule (first_path_rel 0, weighted
    maximum) the stronger echo wins and the frame starts `delay` samples late."""
    import dabgpu
    g = 10 ** (echo_db / 20.0)
    x, starts, e = multipath_capture(70 + delay, [(0, 1.0), (delay, g * np.exp(0.9j))], 22.0, 1.6, 40000)
    frames, counts = gctx.acquire(x[None, :], 4)
    assert counts[0] == len(starts) == 2
    for j in range(2):
        r = O.acquire_candidate(x, O.null_search(x, max_out=4)[j], margin=64)
        gfr = frames[0, j]
        assert (gfr["start"], gfr["coarse_carriers"], gfr["flags"]) == (r.start, r.coarse_carriers, r.flags)
        assert gfr["flags"] == 3 and abs(int(gfr["start"]) + 64 - starts[j]) <= 1           # the FIRST path
    # demodulate where the frames lie: the transmitted FIBs
    import torch
    dev = torch.device("cuda", 0)
    d_x = torch.from_numpy(x).to(dev)
    d_fr = torch.from_numpy(frames.view(np.uint8).reshape(-1)).to(dev)
    soft = torch.zeros((4, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev)
    gctx.ofdm_demod_acquired_dev(d_x.data_ptr(), x.size, 1, 4, d_fr.data_ptr(), soft.data_ptr())
    gctx.sync()
    fib, ok = gctx.fic_decode(soft.cpu().numpy()[:2])
    assert ok.all() and (fib == e.fibs[1:3]).all()
    # the reference's rule on the same capture: the strongest (weighted) tap
    ref_rule = dabgpu.acquire_cfg(first_path_rel=0.0)
    fr2, _ = gctx.acquire(x[None, :], 4, ref_rule)
    late = int(fr2[0, 0]["start"]) + 64 - starts[0]
    assert abs(late - (delay if echo_db > 0.5 else 0)) <= 1 or (abs(echo_db) <= 0.5 and abs(late - delay) <= 1)


@pytest.mark.gpu
def test_gpu_acquisition_under_clock_offset_and_fading(gctx):
    """Three streams in one call: +120 ppm, -120 ppm, and 8 Hz fading with an echo.  Every whole frame is found at its
    true first-path position (+-1 sample) and its FIBs decode."""
    import torch
    import dabgpu
    dev = torch.device("cuda", 0)
    caps = [multipath_capture(81, None, 18.0, -2.3, 51000, ppm=120.0), multipath_capture(82, None, 18.0, 0.7, 20000, ppm=-120.0)]
    e3 = synth.Ensemble(seed=83, n_frames=4)
    rng = np.random.default_rng(83)
    x3 = synth.channel(e3.iq().ravel(), snr_db=22.0, cfo=3.1 / 2048, rng=rng, paths=[(0, 1.0), (150, 0.7j)], fading_hz=8.0)[33333:]
    caps.append((x3, np.array([k * FRAME + NULL - 33333.0 for k in range(1, 4)]), e3))
    n = min(c[0].size for c in caps)
    iq = np.stack([c[0][:n] for c in caps])
    frames, counts = gctx.acquire(iq, 4)
    d_x = torch.from_numpy(iq).to(dev)
    d_fr = torch.from_numpy(frames.view(np.uint8).reshape(-1)).to(dev)
    soft = torch.zeros((12, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev)
    gctx.ofdm_demod_acquired_dev(d_x.data_ptr(), n, 3, 4, d_fr.data_ptr(), soft.data_ptr())
    gctx.sync()
    fib, ok = gctx.fic_decode(soft.cpu().numpy())
    for s, (x, starts, e) in enumerate(caps):
        starts = starts[starts + SYMS + 512 <= n]
        assert counts[s] == len(starts) >= 2
        for j in range(counts[s]):
            assert frames[s, j]["flags"] == 3 and abs(int(frames[s, j]["start"]) + 64 - starts[j]) <= 1.5
            assert ok[4 * s + j].all() and (fib[4 * s + j] == e.fibs[1 + j]).all()


@pytest.mark.gpu
def test_gpu_acquire_rejects_noise_and_bad_arguments(gctx):
    import dabgpu
    rng = np.random.default_rng(2)
    noise = (rng.standard_normal((2, 400000)) + 1j * rng.standard_normal((2, 400000))).astype(np.complex64)
    noise[1, 100000:102656] = 0                                          # a gap as long as a null symbol, no PRS after it
    frames, counts = gctx.acquire(noise, 4)
    assert counts[0] == 0 and counts[1] == 1
    assert frames[1, 0]["flags"] & 1 == 0                                # found a dip, but it does not lock
    L = dabgpu.lib()
    assert L.dabgpu_acquire(gctx._h, None, 0, 1, 100, None, 4, None, None) == -1
    bad = dabgpu.acquire_cfg(max_coarse_carriers=5000)
    with pytest.raises(dabgpu.DabGpuError):
        gctx.acquire(noise, 4, bad)
    frames, counts = gctx.acquire(noise[:, :32], 4)                      # shorter than one block
    assert (counts == 0).all()
