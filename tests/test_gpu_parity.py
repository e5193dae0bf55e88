"""Parity proper: the HIP kernels through the C ABI (libdabgpu.so) against the CPU oracle
on identical inputs.  Bars: bit-exact for FIC / MSC / Viterbi bytes and CRC flags; soft bits
within 1 int8 LSB; cf32 intermediates within 1e-4 of the symbol's peak magnitude."""
import numpy as np
import pytest

import dabgpu
from conftest import golden_path
from dabgpu import synth
from oracle import oracle as O
from golden.make_golden import PROFILES

pytestmark = pytest.mark.gpu

CF32_TOL = 1e-4          # relative to max |X| of the symbol
SOFT_TOL = 1             # int8 LSB


def _rx(ensemble_iq, snr, cfo, seed=11):
    rng = np.random.default_rng(seed)
    rx = synth.channel(ensemble_iq.ravel(), snr_db=snr, cfo=cfo, rng=rng).reshape(ensemble_iq.shape)
    return np.ascontiguousarray(rx[:, synth.NB_NULL:])


def test_context_and_device(ctx):
    assert ctx.stream
    ctx.sync()


@pytest.mark.parametrize("cfo", [0.0, 0.37 / 2048, -5.2 / 2048])
def test_fft_stage_matches_oracle_and_numpy(ctx, ensemble_iq, cfo):
    frames = _rx(ensemble_iq, 25.0, cfo)[:2]
    fo = np.full(2, -cfo, np.float32)
    got = ctx.fft_symbols(frames, fo if cfo else None)
    for f in range(2):
        _, spec, _, _ = O.ofdm_demod_frame(frames[f], float(-cfo), want_spectra=True)
        scale = np.abs(spec).max(axis=1, keepdims=True)
        assert (np.abs(got[f] - spec) <= CF32_TOL * scale).all()
    if cfo == 0.0:
        sym = frames[0].reshape(76, 2552)[:, 504:].astype(np.complex128)
        ref = np.fft.fft(sym, axis=1)
        assert np.abs(got[0] - ref).max() <= CF32_TOL * np.abs(ref).max()


@pytest.mark.parametrize("snr,cfo", [(None, 0.0), (20.0, 0.41 / 2048), (8.0, -0.13 / 2048), (14.0, 7.3 / 2048)])
def test_ofdm_soft_bits_within_one_lsb(ctx, ensemble, ensemble_iq, snr, cfo):
    frames = _rx(ensemble_iq, snr, cfo)
    fo = np.full(frames.shape[0], -cfo, np.float32)
    soft, cyc, dq = ctx.ofdm_demod_frames(frames, fo, want_cyc=True, want_dqpsk=True)
    for f in range(frames.shape[0]):
        osoft, _, ocyc, odq = O.ofdm_demod_frame(frames[f], float(fo[f]), want_cyc=True, want_dqpsk=True)
        delta = np.abs(soft[f].astype(np.int32) - osoft.astype(np.int32))
        assert delta.max() <= SOFT_TOL
        if snr is not None:          # noise-free QPSK sits exactly on the 126/127 truncation edge
            assert (delta != 0).mean() < 0.05
        assert np.abs(cyc[f] - ocyc).max() <= 1e-3 * np.abs(ocyc).max()
        scale = np.abs(odq).max(axis=1, keepdims=True)
        assert (np.abs(dq[f] - odq) <= 2 * CF32_TOL * scale).all()
        if snr is None:
            assert ((soft[f] > 0).astype(np.uint8) == ensemble.frame_bits[f]).all()


# Realistic channels: echoes inside the cyclic prefix (incl. a LATE one stronger than the first
# path), a sample-clock offset, slow flat fading, a level step between frames.  Parity = soft bits within 1 LSB of the
# oracle on the same samples; truth = the transmitted FIBs come back at an SNR where the CRC passes.
CHANNELS = {
    "three_taps": dict(snr_db=22.0, paths=[(0, 1.0), (37, 0.5 * np.exp(1.0j)), (180, 0.35 * np.exp(-2.0j))]),
    "late_echo_2dB_stronger": dict(snr_db=24.0, paths=[(0, 1.0), (250, 1.26 * np.exp(0.4j))]),
    "echo_at_the_prefix_edge": dict(snr_db=24.0, paths=[(0, 1.0), (430, 0.4 * np.exp(2.2j))]),
    "fading_5Hz": dict(snr_db=22.0, fading_hz=5.0, rice_k=4.0),
    "sco_50ppm": dict(snr_db=20.0, sco_ppm=50.0),
    "sco_minus_80ppm_with_echo": dict(snr_db=24.0, sco_ppm=-80.0, paths=[(0, 1.0), (90, 0.6 * np.exp(-0.5j))]),
}


@pytest.mark.parametrize("name", sorted(CHANNELS))
def test_ofdm_parity_and_truth_through_realistic_channels(ctx, ensemble, ensemble_iq, name):
    kw = dict(CHANNELS[name])
    rng = np.random.default_rng(sum(map(ord, name)))
    cfo = 0.23 / 2048
    ppm = kw.get("sco_ppm", 0.0)
    n_frames = 4
    rx = synth.channel(ensemble_iq[:n_frames + 1].ravel(), cfo=cfo, rng=rng, **kw)
    # frame f's first PRS sample in the received stream (the resampler stretches positions and delays by 12 samples);
    # the FFT windows are kept 32 samples early, as every caller of the front end does (timing margin)
    half = 12 if ppm else 0
    starts = [int(round((f * synth.NB_FRAME_SAMPLES + synth.NB_NULL) * (1 + ppm * 1e-6))) - half - 32 for f in range(n_frames)]
    frames = np.stack([rx[s:s + 76 * 2552] for s in starts])
    if name == "fading_5Hz":
        frames[2] *= 3.0                                  # and a 9.5 dB level step between frames
    fo = np.full(n_frames, -cfo, np.float32)
    soft, cyc, _ = ctx.ofdm_demod_frames(frames, fo, want_cyc=True)
    for f in range(n_frames):
        osoft, _, ocyc, _ = O.ofdm_demod_frame(frames[f], float(fo[f]), want_cyc=True)
        assert np.abs(soft[f].astype(np.int32) - osoft.astype(np.int32)).max() <= SOFT_TOL
        assert np.abs(cyc[f] - ocyc).max() <= 1e-3 * np.abs(ocyc).max()
    fib, ok = ctx.fic_decode(soft)
    assert ok.all() and (fib == ensemble.fibs[:n_frames]).all()
    # the fine-frequency estimate from the cyclic-prefix correlations survives the channel
    err = np.angle(cyc.astype(np.complex128)).mean() / (2 * np.pi * 2048)
    assert abs(err) * 2048 < 0.03


def test_ofdm_group_splits_agree(built, ensemble_iq):
    """A frame may be cut into 1..75 symbol runs (dabgpu_cfg.ofdm_symbol_runs); results must not depend on the cut."""
    frames = _rx(ensemble_iq, 12.0, 0.2 / 2048)[:2]
    fo = np.full(2, -0.2 / 2048, np.float32)
    outs = []
    for g in (1, 2, 3, 7, 31, 75):
        with dabgpu.Context(device=0, ofdm_symbol_runs=g) as c:
            outs.append(c.ofdm_demod_frames(frames, fo, want_cyc=True))
    for soft, cyc, _ in outs[1:]:
        assert (soft == outs[0][0]).all()
        assert np.array_equal(cyc, outs[0][1])


def test_ofdm_large_batch_run_plan_agrees_with_fixed_cuts(built, ensemble_iq):
    """Above 3072 frames a launch sends whole frames first (one run each, no symbol transformed twice) and cuts only
    the frames behind them (plan_runs): 3072 + 45 frames give the same soft bits and correlations as the same frames cut
    into three runs each."""
    import torch
    dev = torch.device("cuda", 0)
    frames = _rx(ensemble_iq, 14.0, 0.3 / 2048)
    n = 3072 + 45
    base = torch.from_numpy(frames).to(dev)
    iq = base.repeat((n + base.shape[0] - 1) // base.shape[0], 1)[:n].contiguous()
    fo = torch.full((n,), -0.3 / 2048, dtype=torch.float32, device=dev)
    outs = []
    for runs in (0, 3):
        with dabgpu.Context(device=0, max_frames=n, ofdm_symbol_runs=runs) as c:
            soft = torch.zeros((n, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev)
            cyc = torch.zeros((n, 76), dtype=torch.complex64, device=dev)
            torch.cuda.synchronize()                       # (the fills above run on torch's stream, the launch on the context's)
            c.ofdm_demod_frames_dev(iq.data_ptr(), iq.shape[1], n, fo.data_ptr(), soft.data_ptr(), cyc.data_ptr())
            c.sync()
            outs.append((soft, cyc))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    k = frames.shape[0]
    with dabgpu.Context(device=0) as c:
        ref, _, _ = c.ofdm_demod_frames(frames, np.full(k, -0.3 / 2048, np.float32))
    got = outs[0][0].cpu().numpy()
    for f in (0, 1, 3071, 3072, n - 1):
        assert (got[f] == ref[f % k]).all(), f


def test_ofdm_empty_and_bad_arguments(ctx):
    soft, _, _ = ctx.ofdm_demod_frames(np.zeros((0, 76 * 2552), np.complex64))
    assert soft.shape == (0, 230400)
    L = dabgpu.lib()
    assert L.dabgpu_ofdm_demod_frames_dev(ctx._h, None, 196608, 1, None, None, None, None, None) == -1
    assert L.dabgpu_fic_decode_dev(ctx._h, None, 230400, 1, None, None, None) == -1


def test_ofdm_all_zero_input_gives_erasures(ctx):
    soft, _, _ = ctx.ofdm_demod_frames(np.zeros((1, 76 * 2552), np.complex64))
    assert not soft.any()                      # A == 0 -> erased soft bits, as the oracle defines
    assert not O.ofdm_demod_frame(np.zeros(76 * 2552, np.complex64))[0].any()


def test_fic_bit_exact_on_golden_and_noise(ctx):
    d = np.load(golden_path("fic_cases.npz"))
    fib, ok = ctx.fic_decode(d["soft"])
    assert (fib == d["fib"]).all() and (ok == d["ok"]).all()
    rng = np.random.default_rng(4)
    noise = rng.integers(-127, 128, size=(9, 9216), dtype=np.int8)      # ragged count, pure noise
    noise[7] = 0                                                          # all erasures: every ACS ties
    noise[8] = rng.choice(np.array([-127, 127], np.int8), 9216)           # saturated
    fib, ok = ctx.fic_decode(noise)
    for f in range(9):
        ofib, ook = O.fic_decode(noise[f])
        assert (fib[f] == ofib).all() and (ok[f] == ook).all()


def test_fic_strided_input_matches_contiguous(ctx, ensemble, ensemble_iq):
    frames = _rx(ensemble_iq, 10.0, 0.0)
    soft, _, _ = ctx.ofdm_demod_frames(frames)
    fib, ok = ctx.fic_decode(soft)                        # stride 230400
    fib2, ok2 = ctx.fic_decode(soft[:, :9216])            # stride 9216
    assert (fib == fib2).all() and (ok == ok2).all()
    assert ok.all() and (fib == ensemble.fibs).all()      # known answer: transmitted FIBs


@pytest.mark.parametrize("offset,stride", [(0, 9216), (1, 9216), (16, 9217), (3, 230400 + 5)])
def test_fic_from_unaligned_device_buffers(ctx, offset, stride):
    """The small-batch kernel gathers the FIC's bits sixteen at a time when address and stride allow it and byte by byte
    when they do not: the same bytes from any address and any stride (noise, so every bit matters)."""
    import torch
    dev = torch.device("cuda", 0)
    n = 5
    rng = np.random.default_rng(100 + offset + stride)
    noise = rng.integers(-127, 128, size=(n, 9216), dtype=np.int8)
    buf = torch.zeros(offset + n * stride + 64, dtype=torch.int8, device=dev)
    view = torch.as_strided(buf, (n, 9216), (stride, 1), offset)
    view.copy_(torch.from_numpy(noise).to(dev))
    fib = torch.zeros((n, 12, 32), dtype=torch.uint8, device=dev)
    ok = torch.zeros((n, 12), dtype=torch.uint8, device=dev)
    ctx.fic_decode_dev(buf.data_ptr() + offset, stride, n, fib.data_ptr(), ok.data_ptr(), None)
    ctx.sync()
    for f in range(n):
        ofib, ook = O.fic_decode(noise[f])
        assert (fib[f].cpu().numpy() == ofib).all() and (ok[f].cpu().numpy() == ook).all()


@pytest.mark.parametrize("opt,lvl,br", PROFILES)
def test_viterbi_bit_exact_golden(ctx, opt, lvl, br):
    d = np.load(golden_path("viterbi_cases.npz"))
    mask = O.eep_puncture_mask(opt, lvl, br)[0]
    key = "eep%d_%d_%d" % (opt, lvl, br)
    got = ctx.viterbi(d[key + "_punct"], mask)
    assert (got == d[key + "_bytes"]).all()


def test_viterbi_bit_exact_random_batch(ctx):
    rng = np.random.default_rng(8)
    mask, kept, nsteps, _ = O.eep_puncture_mask(0, 3, 48)
    cw = rng.integers(-127, 128, size=(37, kept), dtype=np.int8)          # 37: not a multiple of 4 waves
    cw[::3] = (cw[::3] // 16) * 16                                        # coarse levels -> many exact ties
    got = ctx.viterbi(cw, mask)
    for i in range(cw.shape[0]):
        assert (got[i] == np.packbits(O.viterbi(O.depuncture(cw[i], mask)))).all(), i


def test_viterbi_is_maximum_likelihood_by_exhaustive_search(ctx):
    """16 information bits + 6 tail bits through dabgpu_viterbi: each output reaches the largest correlation any of the
    65 536 codewords reaches with its input (noise; half erased; small integers full of ties) -- and equals the oracle's."""
    k = 16
    msgs = ((np.arange(1 << k)[:, None] >> np.arange(k - 1, -1, -1)) & 1).astype(np.uint8)
    code = (np.stack([O.conv_encode(m) for m in msgs]).astype(np.int8) * 2 - 1)       # [65536][88], +-1
    rng = np.random.default_rng(12)
    n = 48
    soft = rng.integers(-127, 128, (n, code.shape[1])).astype(np.int8)
    soft[1::3][rng.random(soft[1::3].shape) < 0.5] = 0
    soft[2::3] = rng.integers(-2, 3, soft[2::3].shape).astype(np.int8)
    out = ctx.viterbi(soft, np.ones(code.shape[1], np.uint8))                          # [n][2] bytes, MSB first
    corr = soft.astype(np.int32) @ code.T.astype(np.int32)
    for i in range(n):
        assert (np.unpackbits(out[i]) == O.viterbi(soft[i])).all(), i
        assert corr[i, int(out[i, 0]) << 8 | int(out[i, 1])] == corr[i].max(), i


def test_viterbi_unpunctured_mother_code(ctx):
    rng = np.random.default_rng(6)
    nsteps = 6 + 64
    mask = np.ones(4 * nsteps, np.uint8)
    bits = rng.integers(0, 2, size=(5, 64), dtype=np.uint8)
    cw = np.stack([np.where(O.conv_encode(b) > 0, 127, -127) for b in bits]).astype(np.int8)
    assert (ctx.viterbi(cw, mask) == np.packbits(bits, axis=1)).all()


@pytest.mark.parametrize("opt,lvl,br,start", [(0, 3, 64, 0), (0, 2, 16, 5), (1, 4, 32, 849)])
def test_msc_bit_exact_with_history(ctx, opt, lvl, br, start):
    n_frames = 6
    ens = [synth.Ensemble(seed=100 + s, n_frames=n_frames, option=opt, level=lvl, bitrate=br, start_cu=start)
           for s in range(2)]
    rng = np.random.default_rng(3)
    soft = []
    for e in ens:
        rx = synth.channel(e.iq().ravel(), snr_db=9.0, rng=rng).reshape(n_frames, -1)
        soft.append(ctx.ofdm_demod_frames(np.ascontiguousarray(rx[:, synth.NB_NULL:]))[0])
    soft = np.concatenate(soft)                                          # [2*6][230400]
    sc = dabgpu.subchannel(start, br, level=lvl, eep_type=opt)
    nbits = sc.length * 64
    # one call over all 6 frames per stream
    out_all, hist_all = ctx.msc_decode(sc, soft, n_streams=2, want_history=True)
    # the same as two calls of 3 frames with the history carried across
    first = np.concatenate([soft[0:3], soft[6:9]])
    second = np.concatenate([soft[3:6], soft[9:12]])
    out_a, hist_a = ctx.msc_decode(sc, first, n_streams=2, want_history=True)
    out_b, hist_b = ctx.msc_decode(sc, second, n_streams=2, history_in=hist_a, want_history=True)
    assert (np.concatenate([out_a, out_b], axis=1) == out_all).all()
    assert (hist_b == hist_all).all()
    mask = O.eep_puncture_mask(opt, lvl, br)[0]
    for s in range(2):
        cifs = soft[6 * s:6 * s + 6, synth.NB_FIC_BITS:].reshape(24, synth.NB_CIF_BITS)[:, start * 64:start * 64 + nbits]
        padded = np.concatenate([np.zeros((15, nbits), np.int8), cifs])
        for t in range(24):
            want = O.msc_decode_lf(O.time_deinterleave(padded[t:t + 16]), mask, br * 24 + 6)
            assert (out_all[s, t] == want).all(), (s, t)
            if t >= 15:
                assert (want == ens[s].msc_bytes[t - 15]).all()          # known answer
        assert (hist_all[s] == cifs[-15:]).all()


def test_whole_ensemble_multi_subchannel(ctx):
    """SURVEY 8f-2 shape: a multiplex filled with EEP-A and EEP-B subchannels of different rates, decoded in one
    call from device-resident soft bits; every logical frame equals what was transmitted."""
    import torch
    specs = [(0, 3, 64, 0), (0, 2, 48, 48), (1, 2, 32, 96), (0, 4, 128, 117), (0, 1, 8, 181), (0, 3, 96, 193),
             (1, 4, 64, 265), (0, 3, 32, 295), (0, 3, 192, 319), (0, 2, 16, 463)]
    ens = synth.MultiEnsemble(seed=321, specs=specs, n_frames=8)
    assert all(a[3] + s <= 864 for a, s in zip(specs, ens.sizes))
    rng = np.random.default_rng(2)
    rx = synth.channel(ens.iq().ravel(), snr_db=14.0, rng=rng).reshape(8, -1)
    soft, _, _ = ctx.ofdm_demod_frames(np.ascontiguousarray(rx[:, synth.NB_NULL:]))
    dev = torch.device("cuda", 0)
    d_soft = torch.from_numpy(soft).to(dev)
    scs = [dabgpu.subchannel(st, br, level=lv, eep_type=op) for (op, lv, br, st) in specs]
    outs = [torch.zeros((1, 32, br * 3), dtype=torch.uint8, device=dev) for (_, _, br, _) in specs]
    st = torch.cuda.Stream()
    ctx.msc_decode_multi_dev(scs, d_soft.data_ptr(), 230400, 1, 8, None, None, [o.data_ptr() for o in outs],
                             st.cuda_stream)
    st.synchronize()
    for i, o in enumerate(outs):
        got = o.cpu().numpy()[0]
        for t in range(15, 32):
            assert (got[t] == ens.msc_bytes[i][t - 15]).all(), (i, t)
    # overlapping subchannels are refused before anything is enqueued
    bad = [scs[0], dabgpu.subchannel(40, 64, level=3)]
    with pytest.raises(dabgpu.DabGpuError):
        ctx.msc_decode_multi_dev(bad, d_soft.data_ptr(), 230400, 1, 8, None, None, [outs[0].data_ptr()] * 2,
                                 st.cuda_stream)


def test_msc_rejects_bad_profiles(ctx):
    soft = np.zeros((1, 230400), np.int8)
    for bad in (dabgpu.Subchannel(0, 48, 1, 0, 3, 40),        # UEP: 40 kbit/s is not in the protection profile table
                dabgpu.Subchannel(0, 47, 1, 0, 3, 64),        # UEP: wrong size for 64 kbit/s level 3
                dabgpu.Subchannel(0, 48, 0, 0, 5, 64),        # EEP has no level 5
                dabgpu.Subchannel(0, 48, 0, 2, 3, 64)):       # EEP option 2 is reserved
        with pytest.raises(dabgpu.DabGpuError) as e:
            ctx.msc_decode(bad, soft, 1)
        assert e.value.status == -5


def test_full_size_batch_properties(ctx):
    """64 ensembles x 4 frames (BASELINE config 4 shape): every FIB equals what was sent,
    decode is idempotent, and permuting the frames permutes the outputs."""
    E, F = 64, 4
    ens = [synth.Ensemble(seed=500 + u, n_frames=F) for u in range(4)]
    base = [e.iq() for e in ens]
    rng = np.random.default_rng(1)
    iq = np.empty((E * F, 76 * 2552), np.complex64)
    fo = np.empty(E * F, np.float32)
    for s in range(E):
        cfo = (rng.random() * 0.8 - 0.4) / 2048
        rx = synth.channel(base[s % 4].ravel(), snr_db=18.0, cfo=cfo, rng=rng).reshape(F, -1)
        iq[s * F:(s + 1) * F] = rx[:, synth.NB_NULL:]
        fo[s * F:(s + 1) * F] = -cfo
    soft, _, _ = ctx.ofdm_demod_frames(iq, fo)
    fib, ok = ctx.fic_decode(soft)
    assert ok.all()
    for s in range(E):
        assert (fib[s * F:(s + 1) * F] == ens[s % 4].fibs).all()
    soft2, _, _ = ctx.ofdm_demod_frames(iq, fo)
    assert (soft2 == soft).all()                                           # idempotent / deterministic
    perm = rng.permutation(E * F)
    soft3, _, _ = ctx.ofdm_demod_frames(iq[perm], fo[perm])
    assert (soft3 == soft[perm]).all()
    sc = dabgpu.subchannel(0, 64, level=3)
    msc, _ = ctx.msc_decode(sc, soft, n_streams=E)
    for s in range(E):
        assert (msc[s, 15] == ens[s % 4].msc_bytes[0]).all()


def test_smoke_entry_point():
    import __graft_entry__ as g
    g.smoke()


def test_full_size_front_end_launch_against_the_oracle(built):
    """VERDICT r05 item 4 / BASELINE config 4: the 16 384-frame launch the bench times -- dabgpu_ofdm_demod_streams_dev on
    64 ensembles x 256 frames, the library's own run plan (15 360 whole frames, then 1 024 frames cut into runs), the closed
    loop on the device (coarse offset from the first PRS, fine loop settled over four calls) -- held to the oracle frame by
    frame: 72 sampled frames (the first and last of the launch, both sides of every stream boundary nearby, both sides of the
    uncut / cut boundary, a seeded random rest) are demodulated by O.ofdm_demod_frame at the offset the stream's device state
    held when the launch started, and every soft bit agrees within 1 LSB; the frames' FIBs decode to what was sent."""
    import torch
    import bench
    dev = torch.device("cuda", 0)
    E, F, L = 64, 256, synth.NB_FRAME_SAMPLES
    n = E * F
    iq = torch.empty((n, L), dtype=torch.complex64, device=dev)
    soft = torch.zeros((n, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev)
    cfo_true, ens = bench.make_streams(torch, dev, list(range(E)), F, 8, 20.0, iq)
    d_iq = iq.data_ptr() + synth.NB_NULL * 8
    with dabgpu.Context(device=0, max_frames=n) as c:
        torch.cuda.synchronize()
        c.streams_reset(E)
        sync_out = torch.zeros((E, 4), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        c.sync_prs_dev(d_iq, F * L, E, None, 200, sync_out.data_ptr())
        c.sync()
        coarse = sync_out[:, 0].cpu().numpy()
        for s in range(E):
            c.set_stream_offsets(s, coarse=-float(coarse[s]) / 2048.0)
        for _ in range(4):
            c.ofdm_demod_streams_dev(d_iq, L, E, F, 0.9, soft.data_ptr(), None, None)
        c.sync()
        applied = np.array([c.get_stats(s).net_freq_offset for s in range(E)], np.float32)     # what the next launch applies
        assert np.abs(applied + cfo_true).max() * 2048 < 0.02
        soft.zero_()
        torch.cuda.synchronize()
        c.ofdm_demod_streams_dev(d_iq, L, E, F, 0.9, soft.data_ptr(), None, None)
        c.sync()
        # the launch's plan: whole frames while they fill every wave slot of the chip, the rest cut (plan_runs, dabgpu_api.hip)
        slots = 3072
        uncut = n // slots * slots
        assert uncut == 15360
        picks = {0, 1, F - 1, F, n - 1, n - 2, uncut - 2, uncut - 1, uncut, uncut + 1, uncut - F, uncut + F, slots - 1, slots, 2 * slots}
        rng = np.random.default_rng(0xC0F4)
        while len(picks) < 72:
            picks.add(int(rng.integers(0, n)))
        worst, differing = 0, 0
        got_soft = []
        for f in sorted(picks):
            frame = iq[f, synth.NB_NULL:synth.NB_NULL + 76 * 2552].cpu().numpy()
            want, _, _, _ = O.ofdm_demod_frame(frame, float(applied[f // F]))
            got = soft[f].cpu().numpy()
            d = np.abs(got.astype(np.int32) - want.astype(np.int32))
            assert d.max() <= 1, (f, int(d.max()), int((d > 1).sum()))
            worst = max(worst, int(d.max()))
            differing += int((d > 0).sum())
            got_soft.append(got)
        assert differing <= 2e-4 * len(picks) * dabgpu.NB_FRAME_BITS       # (rounding at quantiser boundaries only)
        fib, ok = c.fic_decode(np.stack(got_soft))
        assert ok.all()
        for k, f in enumerate(sorted(picks)):
            assert (fib[k] == ens[f // F].fibs[(f % F) % 4]).all(), f
