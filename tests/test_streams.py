"""Closed-loop front end (dabgpu_ofdm_demod_streams*, dabgpu_get_stats): per-stream frequency state kept on the
device, updated from the cyclic-prefix correlations, consumed by the next call -- no host-supplied offset
(/root/reference/src/render_radio_block.cpp:202-207, :216).  Also the host-pointer dabgpu_decode_frames."""
import numpy as np
import pytest

import dabgpu
from conftest import make_ctx
from dabgpu import synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _rx(iq, snr, cfo, seed=0):
    rng = np.random.default_rng(seed)
    rx = synth.channel(iq.ravel(), snr_db=snr, cfo=cfo, rng=rng).reshape(iq.shape)
    return np.ascontiguousarray(rx[:, synth.NB_NULL:])


def _state0():
    return dict(fine_freq_offset=np.float32(0), coarse_freq_offset=np.float32(0), signal_average=np.float32(0),
                total_frames_read=0, total_frames_desync=0)


def test_converges_from_a_wrong_offset_within_three_frames(built, ensemble, ensemble_iq):
    """Unknown offset of 0.31 carriers, state starts at 0, one frame per call (the plugin's use): the third frame is
    demodulated with the right correction and its FIBs are the transmitted ones."""
    cfo = 0.31 / 2048
    frames = _rx(ensemble_iq, 15.0, cfo)
    c = make_ctx(None, 4)
    c.streams_reset(1)
    net = []
    for f in range(4):
        soft, _ = c.ofdm_demod_streams(frames[f:f + 1], 1, beta=0.9)
        st = c.get_stats(0)
        net.append(st.net_freq_offset)
        assert st.total_frames_read == f + 1 and st.state == 4
    assert abs(net[2] + cfo) * 2048 < 0.01, [n * 2048 for n in net]          # residual < 1 % of a carrier after 3 frames
    fib, ok = c.fic_decode(soft)
    assert ok.all() and (fib[0] == ensemble.fibs[3]).all()
    assert st.signal_average > 0 and st.total_frames_desync == 0
    c.close()


def test_state_update_matches_the_restatement_and_drives_the_demodulator(built, ensemble_iq):
    """Batch form: 2 streams x 2 frames per call, different offsets per stream.  The state after every call equals
    the oracle's restatement applied to the call's own cyc output, and the soft bits equal those of the
    open-loop call given the state's offset explicitly."""
    cfos = [0.12 / 2048, -0.27 / 2048]
    rx = [_rx(ensemble_iq[:4], 18.0, cfo, seed=k) for k, cfo in enumerate(cfos)]
    c = make_ctx(None, 8)
    c.streams_reset(2)
    c.set_stream_offsets(1, fine=0.2 / 2048)                  # stream 1 starts from a guess, stream 0 from zero
    states = [_state0(), _state0()]
    states[1]["fine_freq_offset"] = np.float32(0.2 / 2048)
    for call in range(2):
        batch = np.concatenate([rx[0][2 * call:2 * call + 2], rx[1][2 * call:2 * call + 2]])
        soft, cyc = c.ofdm_demod_streams(batch, 2, beta=0.75, want_cyc=True)
        fo = np.repeat([states[0]["fine_freq_offset"], states[1]["fine_freq_offset"]], 2).astype(np.float32)
        ref, ref_cyc, _ = c.ofdm_demod_frames(batch, fo, want_cyc=True)
        assert (soft == ref).all() and np.array_equal(cyc, ref_cyc)
        for s in range(2):
            states[s] = O.stream_update(states[s], cyc[2 * s:2 * s + 2], batch[2 * s + 1], 0.75)
            st = c.get_stats(s)
            assert abs(st.fine_freq_offset - states[s]["fine_freq_offset"]) < 2e-9, (call, s)
            assert abs(st.last_fine_error - states[s]["last_fine_error"]) < 2e-9
            assert abs(st.signal_average - states[s]["signal_average"]) < 1e-5 * states[s]["signal_average"]
            assert st.total_frames_read == states[s]["total_frames_read"] == 2 * (call + 1)
            assert st.net_freq_offset == np.float32(st.fine_freq_offset + st.coarse_freq_offset)
    # both loops have closed in on their stream's offset
    for s in range(2):
        assert abs(c.get_stats(s).net_freq_offset + cfos[s]) * 2048 < 0.03
    c.close()


def test_coarse_offset_and_level_drop(built, ensemble_iq):
    """A whole-carrier offset stored with dabgpu_set_stream_offsets is applied on top of the fine one; a frame whose
    level collapses is counted as a desync and leaves the level average alone."""
    cfo = (3 + 0.05) / 2048
    frames = _rx(ensemble_iq[:3], 20.0, cfo)
    c = make_ctx(None, 4)
    c.streams_reset(1)
    c.set_stream_offsets(0, coarse=-3 / 2048)
    soft, _ = c.ofdm_demod_streams(frames[0:1], 1, beta=1.0)
    soft, _ = c.ofdm_demod_streams(frames[1:2], 1, beta=1.0)
    fib, ok = c.fic_decode(soft)
    assert ok.all()
    st = c.get_stats(0)
    assert st.coarse_freq_offset == np.float32(-3 / 2048) and abs(st.net_freq_offset + cfo) * 2048 < 0.02
    level = st.signal_average
    c.ofdm_demod_streams(frames[2:3] * np.complex64(0.01), 1, beta=0.0)
    st = c.get_stats(0)
    assert st.total_frames_desync == 1 and st.total_frames_read == 2 and st.signal_average == level
    c.close()


def test_stream_call_argument_checks(built):
    c = make_ctx(None, 4)
    L = dabgpu.lib()
    iq = np.zeros((1, 76 * 2552), np.complex64)
    soft = np.zeros((1, 230400), np.int8)
    st = dabgpu.Stats()
    import ctypes as C
    assert L.dabgpu_get_stats(c._h, 0, C.byref(st)) == -1                     # no states yet
    assert L.dabgpu_ofdm_demod_streams(c._h, iq.ctypes.data, iq.shape[1], 1, 1, 0.5, soft.ctypes.data, None, None) == -6
    c.streams_reset(2)
    assert L.dabgpu_get_stats(c._h, 2, C.byref(st)) == -1 and L.dabgpu_get_stats(c._h, 1, None) == -1
    assert L.dabgpu_ofdm_demod_streams(c._h, iq.ctypes.data, iq.shape[1], 1, 1, 1.5, soft.ctypes.data, None, None) == -1
    assert L.dabgpu_set_stream_offsets(c._h, 5, None, None) == -1
    assert c.get_stats(1).total_frames_read == 0 and c.get_stats(1).state == 0
    c.ofdm_demod_streams(np.zeros((0, 76 * 2552), np.complex64), 2)           # empty batch: nothing happens
    assert c.get_stats(0).total_frames_read == 0
    c.close()


@pytest.mark.parametrize("lane_mode", [None, 1])
def test_host_decode_frames_equals_separate_calls(built, ensemble, ensemble_iq, lane_mode):
    """dabgpu_decode_frames (one upload, FIC + every sub-channel, one synchronisation) against dabgpu_fic_decode and
    one dabgpu_msc_decode per sub-channel, with de-interleaver history carried over two calls."""
    frames = _rx(ensemble_iq[:4], 14.0, 0.0)
    c = make_ctx(lane_mode, 8)
    soft, _, _ = c.ofdm_demod_frames(frames)
    scs = [dabgpu.subchannel(ensemble.start_cu, 64, level=3), dabgpu.subchannel(ensemble.start_cu + 200, 32, level=2),
           dabgpu.uep_subchannel(11, 400)]
    hist = [None, None, None]
    hist_ref = [None, None, None]
    for call in range(2):
        part = soft[2 * call:2 * call + 2]
        fib, ok, outs, hos = c.decode_frames(part, 1, scs, history_in=hist, want_history=True)
        rfib, rok = c.fic_decode(part)
        assert (fib == rfib).all() and (ok == rok).all() and ok.all()
        for k, sc in enumerate(scs):
            ro, rh = c.msc_decode(sc, part, n_streams=1, history_in=hist_ref[k], want_history=True)
            assert (outs[k] == ro).all(), (call, k)
            assert (hos[k] == rh).all()
            hist_ref[k] = rh
        hist = hos
    # no sub-channels at all is the FIC alone
    fib, ok, outs, _ = c.decode_frames(soft[:1], 1, [])
    assert ok.all() and outs == []
    c.close()


def test_stream_decoder_keeps_the_deinterleaver_state(built, ensemble, ensemble_iq):
    """dabgpu_decode_stream_frames: frames of one stream fed one at a time, history kept on the device -- the same
    bytes as one dabgpu_decode_frames over all of them; a sub-channel joining later starts from erasures; reset
    forgets everything."""
    frames = _rx(ensemble_iq[:5], 16.0, 0.0)
    c = make_ctx(None, 8)
    soft, _, _ = c.ofdm_demod_frames(frames)
    a, b = dabgpu.subchannel(ensemble.start_cu, 64, level=3), dabgpu.subchannel(ensemble.start_cu + 100, 32, level=2)
    rfib, rok, routs, _ = c.decode_frames(soft, 1, [a, b])
    outs_a, outs_b = [], []
    for f in range(5):
        scs = [a] if f < 2 else [a, b]                   # the second sub-channel is announced from frame 2 on
        fib, ok, outs = c.decode_stream_frames(soft[f:f + 1], scs)
        assert (fib == rfib[f:f + 1]).all() and (ok == rok[f:f + 1]).all()
        outs_a.append(outs[0][0])
        if f >= 2:
            outs_b.append(outs[1][0])
    assert (np.concatenate(outs_a) == routs[0][0]).all()
    late, _, louts, _ = c.decode_frames(soft[2:], 1, [b])        # b as if the stream began at frame 2
    assert (np.concatenate(outs_b) == louts[0][0]).all()
    c.decode_stream_reset()
    _, _, outs = c.decode_stream_frames(soft[4:5], [a])
    fresh = c.decode_frames(soft[4:5], 1, [a])[2][0]
    assert (outs[0] == fresh).all()
    c.close()


def test_stream_decoder_drops_a_ring_that_missed_a_frame(built, ensemble, ensemble_iq):
    """A sub-channel left out of one call starts from erasures when it comes back (its ring no longer continues the
    stream); resizing the front end's stream table in between does not disturb the decoder's rings."""
    frames = _rx(ensemble_iq[:5], 16.0, 0.0)
    c = make_ctx(None, 8)
    soft, _, _ = c.ofdm_demod_frames(frames)
    a, b = dabgpu.subchannel(ensemble.start_cu, 64, level=3), dabgpu.subchannel(ensemble.start_cu + 100, 32, level=2)
    c.streams_reset(1)
    c.decode_stream_frames(soft[0:1], [a, b])
    c.streams_reset(8)                                   # grows the state table: must leave the rings alone
    out1 = c.decode_stream_frames(soft[1:2], [a, b])[2]
    ref = c.decode_frames(soft[0:2], 1, [a, b])[2]
    assert (out1[0][0] == ref[0][0][4:]).all() and (out1[1][0] == ref[1][0][4:]).all()
    c.decode_stream_frames(soft[2:3], [a])               # b misses frame 2
    out3 = c.decode_stream_frames(soft[3:4], [a, b])[2]
    ref_a = c.decode_frames(soft[0:4], 1, [a])[2][0][0]
    ref_b = c.decode_frames(soft[3:4], 1, [b])[2][0][0]  # b as if the stream began at frame 3
    assert (out3[0][0] == ref_a[12:]).all() and (out3[1][0] == ref_b).all()
    c.close()


def test_alloc_frame_buffers_returns_a_usable_pair(built, ensemble, ensemble_iq):
    """dabgpu_alloc_frame_buffers(PLACE_PLAIN): the buffers hold whole frames (null symbol included) and the front end
    run on them gives the soft bits of the host-pointer call; bad arguments are refused."""
    import torch
    L = dabgpu.NB_FRAME_SAMPLES
    rx = synth.channel(ensemble_iq.ravel(), snr_db=18.0, cfo=0.0, rng=np.random.default_rng(5)).reshape(ensemble_iq.shape)
    n = rx.shape[0]
    c = make_ctx(None, 8)
    ref, _, _ = c.ofdm_demod_frames(np.ascontiguousarray(rx[:, synth.NB_NULL:]))
    d_iq, d_soft, rep = c.alloc_frame_buffers(n, L, dabgpu.PLACE_PLAIN)
    assert d_iq and d_soft and rep.method == 0 and rep.fallback_reason == 0
    assert rep.setup_peak_bytes == n * (L * 8 + dabgpu.NB_FRAME_BITS)
    dev = torch.device("cuda", 0)
    iq = dabgpu.device_tensor(torch, d_iq, (n, L), torch.complex64, dev)
    soft = dabgpu.device_tensor(torch, d_soft, (n, dabgpu.NB_FRAME_BITS), torch.int8, dev)
    assert iq.data_ptr() == d_iq and soft.data_ptr() == d_soft
    iq.copy_(torch.from_numpy(rx.astype(np.complex64)).to(dev))
    torch.cuda.synchronize()
    c.ofdm_demod_frames_dev(d_iq + synth.NB_NULL * 8, L, n, None, d_soft)
    c.sync()
    assert (soft.cpu().numpy() == ref).all()
    # the geometry mover runs on the same pair (and leaves meaningless bytes in the soft-bit buffer)
    c.mover_frames_dev(d_iq + synth.NB_NULL * 8, L, n, d_soft)
    c.mover_frames_dev(d_iq + synth.NB_NULL * 8, L, n, d_soft, with_prefixes=True)
    c.sync()
    del iq, soft
    c.free_frame_buffers(d_iq, d_soft)
    # a request too small for the domains to matter is a plain pair whatever was asked for, and says why
    d_iq, d_soft, rep = c.alloc_frame_buffers(n, L, dabgpu.PLACE_DOMAINS)
    assert rep.method == 0 and rep.fallback_reason == 1
    c.free_frame_buffers(d_iq, d_soft)
    for bad in (dict(n_frames=0), dict(placement=2), dict(placement=-1), dict(frame_stride=L - 2), dict(frame_stride=L + 1)):
        kw = dict(n_frames=n, frame_stride=L, placement=0)
        kw.update(bad)
        with pytest.raises(dabgpu.DabGpuError):
            c.alloc_frame_buffers(kw["n_frames"], kw["frame_stride"], kw["placement"])
    c.close()


def test_decision_directed_front_end_and_loop(built, ensemble, ensemble_iq):
    """dabgpu_ofdm_demod_frames_dd_dev: the front end without the cyclic prefixes (never read) -- the same soft bits as the
    call that produces the correlations, and dd4 sums equal to the oracle's; the stream call with d_cyc == NULL runs its
    fine loop on them: from 0 against an unknown 0.07-carrier offset it settles like the correlation loop does, its state
    after every call equals oracle.stream_update(dd=True) on the oracle's own sums, and the frames decode."""
    import torch
    from oracle import oracle as O
    L = dabgpu.NB_FRAME_SAMPLES
    dev = torch.device("cuda", 0)
    cfo = 0.07 / 2048
    rx = synth.channel(ensemble_iq.ravel(), snr_db=14.0, cfo=cfo, rng=np.random.default_rng(8),
                       paths=[(0, 1.0), (70, 0.5j)]).reshape(ensemble_iq.shape)
    frames = np.ascontiguousarray(rx[:4, synth.NB_NULL - 16:synth.NB_NULL - 16 + 76 * 2552])
    n = frames.shape[0]
    c = make_ctx(None, 8)
    d_iq = torch.from_numpy(frames).to(dev)
    soft = torch.zeros((n, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev)
    dd4 = torch.zeros((n, 76), dtype=torch.complex64, device=dev)
    fo = torch.full((n,), -0.02 / 2048, dtype=torch.float32, device=dev)          # a residual of 0.05 carriers is left
    torch.cuda.synchronize()
    c.ofdm_demod_frames_dd_dev(d_iq.data_ptr(), frames.shape[1], n, fo.data_ptr(), soft.data_ptr(), dd4.data_ptr())
    c.sync()
    ref_soft, ref_cyc, _ = c.ofdm_demod_frames(frames, np.full(n, -0.02 / 2048, np.float32), want_cyc=True)
    assert (soft.cpu().numpy() == ref_soft).all()
    got = dd4.cpu().numpy()
    for f in range(n):
        osoft, odd = O.ofdm_demod_frame_dd(frames[f], -0.02 / 2048)
        # a run's total sits in the entry of its last symbol, zeros in its others (four frames on an idle chip are
        # cut into runs of one symbol): the frame's sum is the contract
        assert abs(got[f, 1:].sum() - odd[1:].sum()) <= 1e-4 * abs(odd[1:].sum())
        assert got[f, 75] != 0
        assert abs(got[f, 0] - odd[0]) <= 1e-3 * abs(odd[0])          # entry 0: the PRS's cyclic-prefix correlation
        assert np.abs(soft[f].cpu().numpy().astype(np.int32) - osoft.astype(np.int32)).max() <= 1
    # samples at 16-bit scale: same soft bits, finite sums, same estimate (a plain (X conj X)^4 would overflow)
    d_big = d_iq * 30000.0
    soft_b = torch.zeros_like(soft); dd4_b = torch.zeros_like(dd4)
    torch.cuda.synchronize()
    c.ofdm_demod_frames_dd_dev(d_big.data_ptr(), frames.shape[1], n, fo.data_ptr(), soft_b.data_ptr(), dd4_b.data_ptr())
    c.sync()
    big = dd4_b.cpu().numpy()
    assert np.isfinite(big.view(np.float32)).all() and (np.abs(soft_b.cpu().numpy().astype(np.int32) - ref_soft) <= 1).all()
    assert abs(float(O.dd_error(big)) - float(O.dd_error(got))) * 2048 < 1e-4
    # both estimators see the residual 0.07 - 0.02 = 0.05 carriers
    e_dd = float(O.dd_error(got)) * 2048
    e_cp = float(np.angle(ref_cyc.astype(np.complex128)).mean() / (2 * np.pi * 2048)) * 2048
    assert abs(e_dd - 0.05) < 0.004 and abs(e_cp - 0.05) < 0.004 and abs(e_dd - e_cp) < 0.004
    # the closed loop on it
    c.streams_reset(1)
    c.set_stream_loop(decision_directed=True)
    state = {"fine_freq_offset": np.float32(0), "coarse_freq_offset": np.float32(0), "signal_average": np.float32(0),
             "total_frames_read": 0, "total_frames_desync": 0}
    for k in range(4):
        torch.cuda.synchronize()
        c.ofdm_demod_streams_dev(d_iq.data_ptr(), frames.shape[1], 1, n, 0.9, soft.data_ptr(), None, None)
        c.sync()
        odd = np.stack([O.ofdm_demod_frame_dd(frames[f], float(state["fine_freq_offset"]))[1] for f in range(n)])
        state = O.stream_update(state, odd, frames[n - 1], 0.9, dd=True)
        st = c.get_stats(0)
        assert abs(st.fine_freq_offset - state["fine_freq_offset"]) <= 2e-9, k
        assert st.total_frames_read == state["total_frames_read"] and st.total_frames_desync == 0
    assert abs(st.fine_freq_offset * 2048 + 0.07) < 0.003
    fib, ok = c.fic_decode(soft.cpu().numpy())
    assert ok.all() and (fib == ensemble.fibs[:n]).all()
    # ... and from nothing against an offset far outside the fourth power's own +-0.1 carrier: the PRS prefix picks the branch
    rx2 = synth.channel(ensemble_iq.ravel(), snr_db=14.0, cfo=-0.37 / 2048, rng=np.random.default_rng(9)).reshape(ensemble_iq.shape)
    frames2 = np.ascontiguousarray(rx2[:4, synth.NB_NULL - 16:synth.NB_NULL - 16 + 76 * 2552])
    d_iq2 = torch.from_numpy(frames2).to(dev)
    c.streams_reset(1)
    state = {"fine_freq_offset": np.float32(0), "coarse_freq_offset": np.float32(0), "signal_average": np.float32(0),
             "total_frames_read": 0, "total_frames_desync": 0}
    for k in range(3):
        torch.cuda.synchronize()
        c.ofdm_demod_streams_dev(d_iq2.data_ptr(), frames2.shape[1], 1, n, 0.9, soft.data_ptr(), None, None)
        c.sync()
        odd = np.stack([O.ofdm_demod_frame_dd(frames2[f], float(state["fine_freq_offset"]))[1] for f in range(n)])
        state = O.stream_update(state, odd, frames2[n - 1], 0.9, dd=True)
        st = c.get_stats(0)
        assert abs(st.fine_freq_offset - state["fine_freq_offset"]) <= 2e-9, k
    assert abs(st.fine_freq_offset * 2048 - 0.37) < 0.003
    fib, ok = c.fic_decode(soft.cpu().numpy())
    assert ok.all() and (fib == ensemble.fibs[:n]).all()
    c.close()


def test_alloc_frame_buffers_by_domain(built, ensemble_iq):
    """dabgpu_alloc_frame_buffers(PLACE_DOMAINS) on a 5 GiB pair: chunks through the virtual-memory API inside the
    context's own address ranges (reserved once), never more than 1.5 x the pair held, the buffers behave like any device memory (torch
    tensors over them, the front end run on them gives the host-pointer call's soft bits), a second request while the
    pair is alive is a plain pair, releasing gives the memory back, and the same ranges serve the next request."""
    import torch
    L = dabgpu.NB_FRAME_SAMPLES
    dev = torch.device("cuda", 0)
    c = make_ctx(None, 8)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    n = 3000                                                   # 4.4 GiB of samples, 0.64 GiB of soft bits
    final = n * (L * 8 + dabgpu.NB_FRAME_BITS)
    d_iq, d_soft, rep = c.alloc_frame_buffers(n, L, dabgpu.PLACE_DOMAINS)
    # a domain-aware pair -- or, on the 2-9 % of boxes whose virtual-memory chunks all come from one HBM domain, the plain
    # pair the allocator falls back to by itself (test_one_domain_box_gets_a_plain_pair drives that branch everywhere)
    placed = rep.method == 1
    assert placed or rep.fallback_reason == dabgpu.PLAIN_ONE_DOMAIN, (rep.method, rep.fallback_reason, rep.runtime_error)
    assert rep.n_chunks >= 8 and rep.chunk_bytes == 1 << 30
    doms = rep.domains.decode()
    assert len(doms) == min(95, rep.n_chunks) and set(doms.upper()) <= set("ABC") and 1 <= rep.n_domains <= 3
    n_big = len(doms) - len(doms.lstrip("ABC"))
    assert n_big >= 5 and doms[n_big:].islower() and len(doms) - n_big >= 3
    assert rep.setup_peak_bytes == (n_big << 30) + ((len(doms) - n_big) << 28)
    assert rep.setup_peak_bytes <= 1.5 * final                       # never more than 1.5 x the buffers during set-up
    assert rep.classify_ms > 0 and 0 <= rep.conflicts <= 1000
    if placed:
        assert 5 <= rep.iq_chunks <= rep.n_chunks - 1 and 1 <= rep.soft_chunks <= 3
        assert len(rep.iq_map.decode()) == rep.iq_chunks and len(rep.soft_map.decode()) == rep.soft_chunks
        if rep.conflicts == 0 and rep.n_domains > 1:
            assert not set(rep.soft_map.decode().upper()) & set(rep.iq_map.decode().upper()[:2])
        # the mover agrees (the strict bound is back): a domain-aware pair is only ever handed out when writing
        # the soft-bit buffer beside reads of the samples is faster than writing into the samples' own buffer beside the
        # same reads; at 0.985 and above the allocator gives the pair back itself
        assert 0.5 < rep.pair_over_same_domain < 0.985, rep.pair_over_same_domain
    else:
        assert rep.pair_over_same_domain >= 0.985 and rep.iq_chunks == 0 and rep.soft_chunks == 0
    held = free0 - torch.cuda.mem_get_info()[0]
    if placed:
        assert final <= held <= final + (2 << 30) + (64 << 20), held   # the chunks nobody took were released (whole chunks are kept)
    else:
        assert final <= held <= final + (64 << 20), held
    # one domain-aware pair per context at a time: the second request is a plain pair and says why
    e_iq, e_soft, rep2 = c.alloc_frame_buffers(16, L, dabgpu.PLACE_DOMAINS)
    assert rep2.method == 0 and rep2.fallback_reason in (1, 4)
    c.free_frame_buffers(e_iq, e_soft)
    rx = synth.channel(ensemble_iq.ravel(), snr_db=18.0, cfo=0.0, rng=np.random.default_rng(6)).reshape(ensemble_iq.shape)
    k = rx.shape[0]
    ref, _, _ = c.ofdm_demod_frames(np.ascontiguousarray(rx[:, synth.NB_NULL:]))
    iq = dabgpu.device_tensor(torch, d_iq, (n, L), torch.complex64, dev)
    soft = dabgpu.device_tensor(torch, d_soft, (n, dabgpu.NB_FRAME_BITS), torch.int8, dev)
    # frames straddling the 1 GiB chunk boundaries: frame 682 starts below 1 GiB and ends above it
    for base in (0, 680, n - k):
        iq[base:base + k].copy_(torch.from_numpy(rx.astype(np.complex64)).to(dev))
    torch.cuda.synchronize()
    c.ofdm_demod_frames_dev(d_iq + synth.NB_NULL * 8, L, n, None, d_soft)
    c.sync()
    for base in (0, 680, n - k):
        assert (soft[base:base + k].cpu().numpy() == ref).all(), base
    del iq, soft
    if placed:
        with pytest.raises(dabgpu.DabGpuError):               # the two buffers of a mapped pair go back together
            c.free_frame_buffers(d_iq, None)
    c.free_frame_buffers(d_iq, d_soft)
    torch.cuda.synchronize()
    assert free0 - torch.cuda.mem_get_info()[0] <= 64 << 20
    # the same ranges serve the next pair (same addresses); a larger one than they were reserved for is a plain pair
    f_iq, f_soft, rep3 = c.alloc_frame_buffers(n, L, dabgpu.PLACE_DOMAINS)
    placed3 = rep3.method == 1
    # (on a box that hands out one domain the check can read either side of 0.985 from one allocation to the next: each
    # request is judged on its own; two placed pairs of one context share the ranges, hence the address)
    assert placed3 or rep3.fallback_reason == dabgpu.PLAIN_ONE_DOMAIN, \
        (rep3.method, rep3.fallback_reason, rep3.runtime_error, rep3.pair_over_same_domain, rep3.domains, rep3.iq_map, rep3.soft_map)
    assert not (placed and placed3) or f_iq == d_iq
    g_iq, g_soft, rep4 = c.alloc_frame_buffers(2 * n, L, dabgpu.PLACE_DOMAINS)
    assert rep4.method == 0 and rep4.fallback_reason == (4 if placed3 else 5)
    c.free_frame_buffers(g_iq, g_soft)
    c.free_frame_buffers(f_iq, f_soft)
    g_iq, g_soft, rep4 = c.alloc_frame_buffers(2 * n, L, dabgpu.PLACE_DOMAINS)
    assert rep4.method == 0 and rep4.fallback_reason == 5
    c.free_frame_buffers(g_iq, g_soft)
    with pytest.raises(dabgpu.DabGpuError):
        c.alloc_frame_buffers(0, L, dabgpu.PLACE_DOMAINS)
    c.close()
    torch.cuda.synchronize()
    assert free0 - torch.cuda.mem_get_info()[0] <= 64 << 20


def test_one_domain_box_gets_a_plain_pair(built, ensemble_iq):
    """the placed-versus-plain decision is the LIBRARY's.  When the allocator's own check of a placed pair
    reads >= 0.985 (a box whose virtual-memory chunks share one HBM domain: both buffers in one domain is the worst case)
    dabgpu_alloc_frame_buffers gives the pair back and hands out two plain allocations, and the report says so.  The branch is
    taken on 2-9 % of boxes; DABGPU_FLAG_TEST_ONE_DOMAIN makes the check read 1.00 so that it runs on every box.  The buffers
    work like any others, nothing stays mapped in the context's ranges, and the next request goes through the same ranges."""
    import torch
    L = dabgpu.NB_FRAME_SAMPLES
    dev = torch.device("cuda", 0)
    c = dabgpu.Context(device=0, max_frames=8, flags=dabgpu.FLAG_TEST_ONE_DOMAIN)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    n = 3000
    final = n * (L * 8 + dabgpu.NB_FRAME_BITS)
    for round_ in range(2):
        d_iq, d_soft, rep = c.alloc_frame_buffers(n, L, dabgpu.PLACE_DOMAINS)
        assert rep.method == 0 and rep.fallback_reason == dabgpu.PLAIN_ONE_DOMAIN, (rep.method, rep.fallback_reason, rep.runtime_error)
        assert rep.pair_over_same_domain == 1.0 and rep.runtime_error == 0
        assert rep.n_chunks >= 8 and 1 <= rep.n_domains <= 3 and len(rep.domains.decode()) == rep.n_chunks     # what was seen stays in the report
        assert rep.iq_chunks == 0 and rep.soft_chunks == 0 and rep.iq_map == b"" and rep.soft_map == b""
        assert final <= rep.setup_peak_bytes <= 1.5 * final
        held = free0 - torch.cuda.mem_get_info()[0]
        assert final <= held <= final + (64 << 20), held      # every chunk of the pair that was given back is released
        rx = synth.channel(ensemble_iq.ravel(), snr_db=18.0, cfo=0.0, rng=np.random.default_rng(6)).reshape(ensemble_iq.shape)
        k = rx.shape[0]
        ref, _, _ = c.ofdm_demod_frames(np.ascontiguousarray(rx[:, synth.NB_NULL:]))
        iq = dabgpu.device_tensor(torch, d_iq, (n, L), torch.complex64, dev)
        soft = dabgpu.device_tensor(torch, d_soft, (n, dabgpu.NB_FRAME_BITS), torch.int8, dev)
        iq[n - k:].copy_(torch.from_numpy(rx.astype(np.complex64)).to(dev))
        torch.cuda.synchronize()
        c.ofdm_demod_frames_dev(d_iq + (n - k) * L * 8 + synth.NB_NULL * 8, L, k, None, d_soft + (n - k) * dabgpu.NB_FRAME_BITS)
        c.sync()
        assert (soft[n - k:].cpu().numpy() == ref).all()
        del iq, soft
        c.free_frame_buffers(d_iq, d_soft)
        torch.cuda.synchronize()
        assert free0 - torch.cuda.mem_get_info()[0] <= 64 << 20
    c.close()
    torch.cuda.synchronize()
    assert free0 - torch.cuda.mem_get_info()[0] <= 64 << 20


@pytest.mark.parametrize("caller_stream", [False, True])
def test_dev_calls_issued_back_to_back_without_a_host_sync(built, ensemble, ensemble_iq, caller_stream):
    """the library's own ordering between calls must hold with NOTHING from the host in between.  Six stream
    calls and six decode calls (state carried from call to call in HBM, the loop input in the context's shared scratch slot,
    the de-interleaver history ping-ponged, batch sizes that grow so the scratch is re-allocated while earlier launches are
    still in flight) are enqueued in one go -- on the context's own stream, or on a caller stream -- then the accessors
    are called at once (dabgpu_get_stats waits for the last state use on its own).  Everything must equal the same sequence
    run on a second context with a host synchronisation after every call."""
    import torch
    dev = torch.device("cuda", 0)
    cfo = 0.23 / 2048
    rx = _rx(ensemble_iq[:4], 16.0, cfo)
    sizes = [1, 2, 4, 4, 2, 4]                                              # frames per call, one stream
    sc = dabgpu.subchannel(0, 64, level=3)
    d_rx = torch.from_numpy(rx).to(dev)
    L = rx.shape[1]

    def run(c, sync_each, stream):
        c.streams_reset(1)
        softs = [torch.zeros((n, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev) for n in sizes]
        fibs = [torch.zeros((n, 12, 32), dtype=torch.uint8, device=dev) for n in sizes]
        oks = [torch.zeros((n, 12), dtype=torch.uint8, device=dev) for n in sizes]
        mscs = [torch.zeros((1, 4 * n, 192), dtype=torch.uint8, device=dev) for n in sizes]
        hist = [torch.zeros((1, 15, sc.length * 64), dtype=torch.int8, device=dev) for _ in range(2)]
        torch.cuda.synchronize()                                            # the HARNESS's buffers exist; from here on: no sync
        for k, n in enumerate(sizes):
            c.ofdm_demod_streams_dev(d_rx.data_ptr(), L, 1, n, 0.8, softs[k].data_ptr(), None, None, stream)
            if sync_each:
                torch.cuda.synchronize()
            c.decode_frames_dev(softs[k].data_ptr(), dabgpu.NB_FRAME_BITS, 1, n, fibs[k].data_ptr(), oks[k].data_ptr(), [sc],
                                [hist[k & 1].data_ptr()], [hist[(k & 1) ^ 1].data_ptr()], [mscs[k].data_ptr()], stream)
            if sync_each:
                torch.cuda.synchronize()
        st = c.get_stats(0)                                                 # waits for the last state use by itself
        torch.cuda.synchronize()
        return st, [t.cpu().numpy() for t in softs + fibs + oks + mscs]

    a, b = dabgpu.Context(device=0, max_frames=1), dabgpu.Context(device=0, max_frames=1)    # (plain contexts: no harness ordering at all)
    s = torch.cuda.Stream(device=dev) if caller_stream else None
    st_a, out_a = run(a, False, s.cuda_stream if s else None)
    st_b, out_b = run(b, True, None)
    assert st_a.total_frames_read == st_b.total_frames_read == sum(sizes)
    assert st_a.fine_freq_offset == st_b.fine_freq_offset and st_a.signal_average == st_b.signal_average
    assert abs(st_a.net_freq_offset + cfo) * 2048 < 0.02
    for x, y in zip(out_a, out_b):
        assert np.array_equal(x, y)
    n = len(sizes)
    assert out_a[2 * n + n - 1].all() and (out_a[n + n - 1][0] == ensemble.fibs[0]).all()      # the last call's FIBs: CRC-clean, the transmitted ones
    a.close(); b.close()


def test_stream_decoder_refuses_a_subchannel_listed_twice_and_keeps_its_state(built, ensemble, ensemble_iq):
    """the kept de-interleaver rings are keyed by (start, size) -- a sub-channel listed twice (or two that
    overlap) would share one ring and flip it twice, and the NEXT call would continue from a stale buffer.  Such a call is
    refused with DABGPU_ERR_ARG before any ring is touched: the stream continues as if it had not been made."""
    frames = _rx(ensemble_iq[:3], 16.0, 0.0)
    c = make_ctx(None, 8)
    soft, _, _ = c.ofdm_demod_frames(frames)
    a = dabgpu.subchannel(ensemble.start_cu, 64, level=3)
    over = dabgpu.subchannel(ensemble.start_cu + 10, 32, level=2)              # overlaps `a`
    ref = c.decode_frames(soft, 1, [a])[2][0][0]
    c.decode_stream_frames(soft[0:1], [a])
    for bad in ([a, a], [a, over]):
        with pytest.raises(dabgpu.DabGpuError) as e:
            c.decode_stream_frames(soft[1:2], bad)
        assert e.value.status == -1
    out1 = c.decode_stream_frames(soft[1:2], [a])[2][0][0]
    out2 = c.decode_stream_frames(soft[2:3], [a])[2][0][0]
    assert (out1 == ref[4:8]).all() and (out2 == ref[8:12]).all()
    c.close()
