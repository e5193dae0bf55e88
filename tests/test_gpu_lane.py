"""The large-batch (codeword-per-lane) Viterbi against the small-batch kernels and the oracle at the batch sizes
where the library switches over by itself (>= 24576 codewords per launch)."""
import numpy as np
import pytest
import torch

import dabgpu
from conftest import make_ctx
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def test_switch_over_is_bit_exact():
    n_streams, fps = 14, 448                                 # 6272 frames -> 25088 FIC and MSC codewords
    n = n_streams * fps
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(5)
    stride = 16384                                           # compact frames: only the FIC part is present
    soft = torch.zeros((n, stride), dtype=torch.int8, device=dev)
    soft[:, :dabgpu.NB_FIC_BITS] = torch.randint(-127, 128, (n, dabgpu.NB_FIC_BITS), dtype=torch.int8, device=dev,
                                                 generator=g)
    outs = []
    for mode in (0, None):
        c = make_ctx(mode, max_frames=64)
        fib = torch.zeros((n, 12, 32), dtype=torch.uint8, device=dev)
        ok = torch.zeros((n, 12), dtype=torch.uint8, device=dev)
        c.fic_decode_dev(soft.data_ptr(), stride, n, fib.data_ptr(), ok.data_ptr(), None)
        c.sync()
        outs.append((fib.cpu().numpy(), ok.cpu().numpy()))
        c.close()
    assert (outs[0][0] == outs[1][0]).all() and (outs[0][1] == outs[1][1]).all()
    # spot-check the oracle on a few codewords from both ends of the batch
    s = soft.cpu().numpy()
    for f in (0, 1, n // 2, n - 1):
        ofib, ook = O.fic_decode(s[f, :dabgpu.NB_FIC_BITS])
        assert (outs[1][0][f] == np.asarray(ofib).reshape(12, 32)).all() and (outs[1][1][f] == ook).all()


def test_switch_over_msc_with_history_and_ragged_streams():
    # 25 streams x 247 frames: 24700 codewords, stream length not a multiple of the 64-codeword groups, so groups
    # straddle streams and the de-interleaver history of each stream is picked up inside a group
    n_streams, fps = 25, 247
    sc = dabgpu.subchannel(3, 32, level=2)                   # 32 kbit/s EEP 2-A
    nbits = sc.length * 64
    n = n_streams * fps
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(9)
    stride = 4 * 55296 + dabgpu.NB_FIC_BITS
    soft = torch.zeros((n, stride), dtype=torch.int8, device=dev)
    view = soft[:, dabgpu.NB_FIC_BITS:].view(n, 4, 55296)[:, :, 3 * 64:3 * 64 + nbits]
    view.copy_(torch.randint(-127, 128, (n, 4, nbits), dtype=torch.int8, device=dev, generator=g))
    hist = torch.randint(-127, 128, (n_streams, 15, nbits), dtype=torch.int8, device=dev, generator=g)
    nbytes = 32 * 3
    res = []
    for mode in (0, None):
        c = make_ctx(mode, max_frames=64)
        out = torch.zeros((n_streams, fps * 4, nbytes), dtype=torch.uint8, device=dev)
        hout = torch.zeros_like(hist)
        c.msc_decode_dev(sc, soft.data_ptr(), stride, n_streams, fps, hist.data_ptr(), hout.data_ptr(), out.data_ptr(), None)
        c.sync()
        res.append((out.cpu().numpy(), hout.cpu().numpy()))
        c.close()
    assert (res[0][0] == res[1][0]).all() and (res[0][1] == res[1][1]).all()
    # oracle on the first CIFs of one stream (history in play) and the last CIF of the last stream
    mask = O.eep_puncture_mask(0, 2, 32)[0]
    s_np = view[:fps * 1].cpu().numpy().reshape(fps * 4, nbits)
    padded = np.concatenate([hist[0].cpu().numpy(), s_np])
    for t in (0, 1, 14, 15, 16):
        want = O.msc_decode_lf(O.time_deinterleave(padded[t:t + 16]), mask, 32 * 24 + 6)
        assert (res[1][0][0, t] == want).all(), t
    last = view[(n_streams - 1) * fps:].cpu().numpy().reshape(fps * 4, nbits)
    want = O.msc_decode_lf(O.time_deinterleave(last[-16:]), mask, 32 * 24 + 6)
    assert (res[1][0][-1, -1] == want).all()


@pytest.mark.parametrize("fps,with_hist", [(16, True), (32, False), (48, True)])
def test_fused_forward_msc_whole_groups(fps, with_hist):
    """Streams whose CIF count is a multiple of 64 take the forward pass that depunctures for itself (history rows
    staged like any other row); same bytes as the wave-per-codeword kernels and as the oracle."""
    n_streams = 3
    sc = dabgpu.subchannel(20, 48, level=2)                  # 48 kbit/s EEP 2-A
    nbits = sc.length * 64
    n = n_streams * fps
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(fps)
    stride = dabgpu.NB_FRAME_BITS
    soft = torch.zeros((n, stride), dtype=torch.int8, device=dev)
    view = soft[:, dabgpu.NB_FIC_BITS:].view(n, 4, 55296)[:, :, 20 * 64:20 * 64 + nbits]
    view.copy_(torch.randint(-127, 128, (n, 4, nbits), dtype=torch.int8, device=dev, generator=g))
    hist = torch.randint(-127, 128, (n_streams, 15, nbits), dtype=torch.int8, device=dev, generator=g)
    res = []
    for mode in (0, 1):
        c = make_ctx(mode, max_frames=8)
        out = torch.zeros((n_streams, fps * 4, 48 * 3), dtype=torch.uint8, device=dev)
        hout = torch.zeros_like(hist)
        c.msc_decode_dev(sc, soft.data_ptr(), stride, n_streams, fps, hist.data_ptr() if with_hist else None,
                         hout.data_ptr(), out.data_ptr(), None)
        c.sync()
        res.append((out.cpu().numpy(), hout.cpu().numpy()))
        c.close()
    assert (res[0][0] == res[1][0]).all() and (res[0][1] == res[1][1]).all()
    mask = O.eep_puncture_mask(0, 2, 48)[0]
    for s in (0, n_streams - 1):
        cifs = view[s * fps:(s + 1) * fps].cpu().numpy().reshape(fps * 4, nbits)
        h = hist[s].cpu().numpy() if with_hist else np.zeros((15, nbits), np.int8)
        padded = np.concatenate([h, cifs])
        for t in (0, 7, 15, 63, fps * 4 - 1):
            want = O.msc_decode_lf(O.time_deinterleave(padded[t:t + 16]), mask, 48 * 24 + 6)
            assert (res[1][0][s, t] == want).all(), (s, t)


def test_grouped_multi_subchannel_launch():
    """dabgpu_msc_decode_multi_dev with whole-group streams: every sub-channel's codewords go through ONE forward and
    ONE traceback launch (entries of different length, EEP-A / EEP-B / UEP); same bytes and history rings as decoding
    them one by one with the wave-per-codeword kernels."""
    n_streams, fps = 2, 16
    scs = [dabgpu.subchannel(0, 64, level=3), dabgpu.subchannel(48, 48, level=2), dabgpu.subchannel(96, 32, level=2, eep_type=1),
           dabgpu.uep_subchannel(17, 117), dabgpu.subchannel(175, 8, level=1), dabgpu.subchannel(187, 128, level=4),
           dabgpu.uep_subchannel(4, 251)]
    n = n_streams * fps
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(77)
    soft = torch.randint(-127, 128, (n, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev, generator=g)
    hin = [torch.randint(-127, 128, (n_streams, 15, sc.length * 64), dtype=torch.int8, device=dev, generator=g) for sc in scs]
    res = []
    for mode in (0, 1):
        c = make_ctx(mode, max_frames=8)
        outs = [torch.zeros((n_streams, fps * 4, sc.bitrate_kbps * 3), dtype=torch.uint8, device=dev) for sc in scs]
        hout = [torch.zeros_like(h) for h in hin]
        c.msc_decode_multi_dev(scs, soft.data_ptr(), dabgpu.NB_FRAME_BITS, n_streams, fps, [h.data_ptr() for h in hin],
                               [h.data_ptr() for h in hout], [o.data_ptr() for o in outs], None)
        c.sync()
        res.append(([o.cpu().numpy() for o in outs], [h.cpu().numpy() for h in hout]))
        c.close()
    for i in range(len(scs)):
        assert (res[0][0][i] == res[1][0][i]).all(), i
        assert (res[0][1][i] == res[1][1][i]).all(), i
    # oracle spot check on the UEP entry (index 3) across the stream start
    mask, kept, nsteps, _ = O.uep_puncture_mask(17)
    sc = scs[3]
    nbits = sc.length * 64
    cifs = soft[:fps, dabgpu.NB_FIC_BITS:].reshape(fps * 4, 55296)[:, sc.start_address * 64:sc.start_address * 64 + nbits].cpu().numpy()
    padded = np.concatenate([hin[3][0].cpu().numpy(), cifs])
    for t in (0, 14, 15, 40, 63):
        want = O.msc_decode_lf(O.time_deinterleave(padded[t:t + 16])[:kept], mask, nsteps)
        assert (res[1][0][3][0, t] == want).all(), t


@pytest.mark.parametrize("mode,fps", [(1, 16), (1, 6), (0, 16), (None, 16)])
def test_decode_frames_equals_separate_calls(mode, fps):
    """dabgpu_decode_frames_dev (FIC + sub-channels, BasicRadio::Process for a batch): grouped launch with the FIC as
    one more entry (mode 1, 16 frames per stream), the part-by-part path for other shapes; always the same bytes as
    dabgpu_fic_decode_dev + dabgpu_msc_decode_dev."""
    n_streams = 3
    scs = [dabgpu.subchannel(0, 64, level=3), dabgpu.uep_subchannel(5, 60), dabgpu.subchannel(100, 32, level=4, eep_type=1)]
    n = n_streams * fps
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(fps)
    soft = torch.randint(-127, 128, (n, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev, generator=g)
    hin = [torch.randint(-127, 128, (n_streams, 15, sc.length * 64), dtype=torch.int8, device=dev, generator=g) for sc in scs]
    def buffers():
        return (torch.zeros((n, 12, 32), dtype=torch.uint8, device=dev), torch.zeros((n, 12), dtype=torch.uint8, device=dev),
                [torch.zeros((n_streams, fps * 4, sc.bitrate_kbps * 3), dtype=torch.uint8, device=dev) for sc in scs],
                [torch.zeros_like(h) for h in hin])
    ref = make_ctx(0, max_frames=8)
    fib0, ok0, out0, hout0 = buffers()
    ref.fic_decode_dev(soft.data_ptr(), dabgpu.NB_FRAME_BITS, n, fib0.data_ptr(), ok0.data_ptr(), None)
    for i, sc in enumerate(scs):
        ref.msc_decode_dev(sc, soft.data_ptr(), dabgpu.NB_FRAME_BITS, n_streams, fps, hin[i].data_ptr(), hout0[i].data_ptr(),
                           out0[i].data_ptr(), None)
    ref.sync()
    c = make_ctx(mode, max_frames=8)
    fib1, ok1, out1, hout1 = buffers()
    c.decode_frames_dev(soft.data_ptr(), dabgpu.NB_FRAME_BITS, n_streams, fps, fib1.data_ptr(), ok1.data_ptr(), scs,
                        [h.data_ptr() for h in hin], [h.data_ptr() for h in hout1], [o.data_ptr() for o in out1], None)
    c.sync()
    assert (fib0 == fib1).all() and (ok0 == ok1).all()
    for i in range(len(scs)):
        assert (out0[i] == out1[i]).all() and (hout0[i] == hout1[i]).all(), i
    # FIC only (no sub-channels) and bad arguments
    fib2, ok2, _, _ = buffers()
    c.decode_frames_dev(soft.data_ptr(), dabgpu.NB_FRAME_BITS, n_streams, fps, fib2.data_ptr(), ok2.data_ptr(), [], None, None, None, None)
    c.sync()
    assert (fib2 == fib0).all()
    with pytest.raises(dabgpu.DabGpuError):
        c.decode_frames_dev(soft.data_ptr(), dabgpu.NB_FRAME_BITS, n_streams, fps, fib2.data_ptr(), ok2.data_ptr(),
                            [scs[0], dabgpu.subchannel(10, 64, level=3)], None, None, [out1[0].data_ptr()] * 2, None)
    ref.close(); c.close()


@pytest.mark.parametrize("n_streams,fps", [(1, 1), (1, 256), (3, 128), (1, 388), (2, 500)])
def test_fic_with_one_sub_channel_equals_separate_calls(n_streams, fps):
    """One ensemble's shape (BASELINE configs 2-3, the plugin's one frame per call): the FIC and ONE sub-channel share a
    launch of the wave-per-codeword kernel while all their codewords are resident at once (<= 384 frames for 64 kbit/s),
    above that they are two launches; on noise, with carried history, always the bytes of the separate calls."""
    sc = dabgpu.subchannel(5, 64, level=3)
    n = n_streams * fps
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(1000 + n)
    soft = torch.randint(-127, 128, (n, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev, generator=g)
    hin = torch.randint(-127, 128, (n_streams, 15, sc.length * 64), dtype=torch.int8, device=dev, generator=g)
    def buffers():
        return (torch.zeros((n, 12, 32), dtype=torch.uint8, device=dev), torch.zeros((n, 12), dtype=torch.uint8, device=dev),
                torch.zeros((n_streams, fps * 4, 192), dtype=torch.uint8, device=dev), torch.zeros_like(hin))
    c = make_ctx(None, max_frames=8)
    fib0, ok0, out0, hout0 = buffers()
    c.fic_decode_dev(soft.data_ptr(), dabgpu.NB_FRAME_BITS, n, fib0.data_ptr(), ok0.data_ptr(), None)
    c.msc_decode_dev(sc, soft.data_ptr(), dabgpu.NB_FRAME_BITS, n_streams, fps, hin.data_ptr(), hout0.data_ptr(), out0.data_ptr(), None)
    fib1, ok1, out1, hout1 = buffers()
    c.decode_frames_dev(soft.data_ptr(), dabgpu.NB_FRAME_BITS, n_streams, fps, fib1.data_ptr(), ok1.data_ptr(), [sc],
                        [hin.data_ptr()], [hout1.data_ptr()], [out1.data_ptr()], None)
    c.sync()
    assert (fib0 == fib1).all() and (ok0 == ok1).all() and (out0 == out1).all() and (hout0 == hout1).all()
    # and the oracle on a few codewords of the first stream
    mask, kept, nsteps = O.eep_puncture_mask(0, 3, 64)[:3]
    cifs = soft[:fps, dabgpu.NB_FIC_BITS:].reshape(fps * 4, 55296)[:, sc.start_address * 64:(sc.start_address + sc.length) * 64].cpu().numpy()
    padded = np.concatenate([hin[0].cpu().numpy(), cifs])
    for t in sorted({0, min(3, fps * 4 - 1), fps * 4 - 1}):
        want = O.msc_decode_lf(O.time_deinterleave(padded[t:t + 16])[:kept], mask, nsteps)
        assert (out1[0, t].cpu().numpy() == want).all(), t
    c.close()


def test_bench_shape_grouped_launch_equals_the_oracle_on_noise():
    """The shape bench.py runs -- 64 streams x 256 frames, FIC + one 64 kbit/s EEP 3-A sub-channel in ONE grouped
    lane launch (65 536 + 65 536 codewords) -- on pure noise, with carried de-interleaver history, against the oracle
    for EVERY codeword (the C oracle decodes the 131 072 codewords in a few seconds)."""
    E, F = 64, 256
    n = E * F
    sc = dabgpu.subchannel(0, 64, level=3)
    nbits = sc.length * 64
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(77)
    soft = torch.zeros((n, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev)
    soft[:, :dabgpu.NB_FIC_BITS] = torch.randint(-127, 128, (n, dabgpu.NB_FIC_BITS), dtype=torch.int8, device=dev, generator=g)
    cifs = soft[:, dabgpu.NB_FIC_BITS:].view(n, 4, 55296)
    cifs[:, :, :nbits] = torch.randint(-127, 128, (n, 4, nbits), dtype=torch.int8, device=dev, generator=g)
    hist = torch.randint(-127, 128, (E, 15, nbits), dtype=torch.int8, device=dev, generator=g)
    hout = torch.zeros_like(hist)
    fib = torch.zeros((n, 12, 32), dtype=torch.uint8, device=dev)
    ok = torch.zeros((n, 12), dtype=torch.uint8, device=dev)
    out = torch.zeros((E, F * 4, 192), dtype=torch.uint8, device=dev)
    c = make_ctx(None, max_frames=64)
    c.set_timing(True)
    c.decode_frames_dev(soft.data_ptr(), dabgpu.NB_FRAME_BITS, E, F, fib.data_ptr(), ok.data_ptr(), [sc], [hist.data_ptr()],
                        [hout.data_ptr()], [out.data_ptr()], None)
    c.sync()
    assert c.mean_kernel_ms(2)[1] == 1                       # one grouped launch took all of it
    fib_h, ok_h, out_h = fib.cpu().numpy(), ok.cpu().numpy(), out.cpu().numpy()
    fic_h = soft[:, :dabgpu.NB_FIC_BITS].cpu().numpy()
    for f in range(n):
        ofib, ook = O.fic_decode(fic_h[f])
        assert (fib_h[f] == np.asarray(ofib).reshape(12, 32)).all() and (ok_h[f] == ook).all(), f
    sub = cifs[:, :, :nbits].cpu().numpy().reshape(E, F * 4, nbits)
    hist_h = hist.cpu().numpy()
    mask = O.eep_puncture_mask(0, 3, 64)[0]
    for s in range(E):
        rows = np.concatenate([hist_h[s], sub[s]])            # CIFs -15 .. 4F-1 of the stream
        for t in range(F * 4):
            de = O.time_deinterleave(rows[t:t + 16])
            assert (out_h[s, t] == O.msc_decode_lf(de, mask, 64 * 24 + 6)).all(), (s, t)
    # and the ring it leaves behind is the stream's last 15 CIFs
    assert (hout.cpu().numpy() == sub[:, -15:]).all()
    c.close()


def test_more_sub_channels_than_one_grouped_launch_holds():
    """30 sub-channels: the grouped launches carry 16 (lane kernels) or 24 (wave kernels) entries, longer lists are cut
    into several launches.  Same bytes and history rings whichever family decodes them, and the same as one call per
    sub-channel; the stream decoder (one frame per call, rings on the device) agrees as well."""
    n_streams, fps = 1, 16
    scs = [dabgpu.subchannel(6 * i, 8, level=3) for i in range(30)]
    assert all(sc.length == 6 for sc in scs)
    n = n_streams * fps
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(4242)
    soft = torch.randint(-127, 128, (n, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev, generator=g)
    res = []
    for mode in (0, 1):
        c = make_ctx(mode, max_frames=8)
        outs = [torch.zeros((n_streams, fps * 4, sc.bitrate_kbps * 3), dtype=torch.uint8, device=dev) for sc in scs]
        hout = [torch.zeros((n_streams, 15, sc.length * 64), dtype=torch.int8, device=dev) for sc in scs]
        c.msc_decode_multi_dev(scs, soft.data_ptr(), dabgpu.NB_FRAME_BITS, n_streams, fps, None,
                               [h.data_ptr() for h in hout], [o.data_ptr() for o in outs], None)
        c.sync()
        res.append(([o.cpu().numpy() for o in outs], [h.cpu().numpy() for h in hout]))
        c.close()
    c = make_ctx(0, max_frames=8)
    soft_h = soft.cpu().numpy()
    for i, sc in enumerate(scs):
        assert (res[0][0][i] == res[1][0][i]).all() and (res[0][1][i] == res[1][1][i]).all(), i
        one, hist = c.msc_decode(sc, soft_h, n_streams, want_history=True)
        assert (one == res[0][0][i]).all() and (hist == res[0][1][i]).all(), i
    streamed = [[] for _ in scs]
    for f in range(fps):
        _, _, outs = c.decode_stream_frames(soft_h[f:f + 1], scs)
        for i in range(len(scs)):
            streamed[i].append(outs[i][0])
    for i in range(len(scs)):
        assert (np.concatenate(streamed[i]) == res[0][0][i][0]).all(), i
    c.close()


def test_timer_splits_the_grouped_lane_decode_into_its_kernels():
    """dabgpu_mean_kernel_ms 4 / 5 / 6 (what `decoder.roofline` in the bench line is built from): while timing is on, the
    grouped codeword-per-lane decode records two events between its kernels, and the three parts -- forward pass,
    traceback, history copy -- add up to the whole call's slot 2 (but for the events' own few microseconds); a context whose
    decode took the wave-per-codeword kernels has no such parts and says so (DABGPU_ERR_ARG), never a stale figure."""
    dev = torch.device("cuda", 0)
    E, F = 4, 16
    n = E * F
    g = torch.Generator(device=dev); g.manual_seed(9)
    soft = torch.randint(-127, 128, (n, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev, generator=g)
    fib = torch.zeros((n, 12, 32), dtype=torch.uint8, device=dev)
    ok = torch.zeros((n, 12), dtype=torch.uint8, device=dev)
    sc = dabgpu.subchannel(0, 64, level=3)
    msc = torch.zeros((E, F * 4, 192), dtype=torch.uint8, device=dev)
    hist = [torch.zeros((E, 15, sc.length * 64), dtype=torch.int8, device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    for mode in (1, 0):
        c = make_ctx(mode, max_frames=n)
        c.set_timing(True)
        for k in range(3):
            c.decode_frames_dev(soft.data_ptr(), dabgpu.NB_FRAME_BITS, E, F, fib.data_ptr(), ok.data_ptr(), [sc],
                                [hist[k & 1].data_ptr()], [hist[(k & 1) ^ 1].data_ptr()], [msc.data_ptr()], None)
        c.sync()
        whole, n_whole = c.mean_kernel_ms(2)
        assert n_whole == 3 and whole > 0
        if mode == 1:
            parts = [c.mean_kernel_ms(w) for w in (4, 5, 6)]
            assert all(cnt == 3 and ms > 0 for ms, cnt in parts)
            fwd, tb, hi = (ms for ms, _ in parts)
            assert hi < tb < fwd and 0.9 * whole <= fwd + tb + hi <= whole * 1.001
        else:
            for w in (4, 5, 6):
                with pytest.raises(dabgpu.DabGpuError):
                    c.mean_kernel_ms(w)
        with pytest.raises(dabgpu.DabGpuError):
            c.mean_kernel_ms(7)
        c.set_timing(False)
        c.close()
