"""Maximum sizes: sub-channels whose codewords are too long for the wave-per-codeword kernels' LDS slab (above
~680 kbit/s, up to a sub-channel that fills the whole CIF) go through the lane kernels, which keep survivors in HBM
and write decoded words straight to the output.  Bit-exact against the oracle on noise, with carried history."""
import numpy as np
import pytest

import dabgpu
from conftest import make_ctx
from oracle import oracle as O

pytestmark = pytest.mark.gpu

NB_FIC_BITS, NB_CIF_BITS = 9216, 55296


@pytest.mark.parametrize("opt,lvl,br", [(0, 3, 1152),      # EEP 3-A filling the CIF: 864 CUs, 27 654 steps
                                        (0, 4, 1728),      # EEP 4-A filling the CIF: 41 478 steps, the longest there is
                                        (0, 2, 768),       # 768 CUs: just above the wave kernels' 16-bit position table
                                        (1, 1, 512)])      # EEP 1-B
def test_over_long_subchannel_is_bit_exact(opt, lvl, br):
    mask, kept, nsteps, cu = O.eep_puncture_mask(opt, lvl, br)
    assert cu <= 864 and nsteps == br * 24 + 6
    sc = dabgpu.subchannel(864 - cu, br, level=lvl, eep_type=opt)
    assert sc.length == cu
    nbits = cu * 64
    rng = np.random.default_rng(br)
    soft = rng.integers(-127, 128, size=(3, dabgpu.NB_FRAME_BITS), dtype=np.int8)
    hist_in = rng.integers(-127, 128, size=(1, 15, nbits), dtype=np.int8)
    for mode in (None, 0):                                    # by batch size, and with the wave family forced
        c = make_ctx(mode, 8)
        out, hist = c.msc_decode(sc, soft, 1, history_in=hist_in, want_history=True)
        # ... and as two calls with the history carried across
        out_a, hist_a = c.msc_decode(sc, soft[:1], 1, history_in=hist_in, want_history=True)
        out_b, hist_b = c.msc_decode(sc, soft[1:], 1, history_in=hist_a, want_history=True)
        c.close()
        assert (np.concatenate([out_a, out_b], axis=1) == out).all() and (hist_b == hist).all()
        cifs = soft[:, NB_FIC_BITS:].reshape(12, NB_CIF_BITS)[:, sc.start_address * 64:sc.start_address * 64 + nbits]
        padded = np.concatenate([hist_in[0], cifs])
        for t in (0, 1, 7, 11):
            want = O.msc_decode_lf(O.time_deinterleave(padded[t:t + 16])[:kept], mask, nsteps)
            assert (out[0, t] == want).all(), (mode, t)
        assert (hist[0] == padded[-15:]).all()


def test_over_long_plain_viterbi():
    """dabgpu_viterbi on an unpunctured 24 006-step code (beyond the LDS slab) equals the oracle on noise."""
    nsteps = 24006
    mask = np.ones(4 * nsteps, np.uint8)
    rng = np.random.default_rng(7)
    punct = rng.integers(-127, 128, size=(3, 4 * nsteps), dtype=np.int8)
    c = make_ctx(None, 8)
    got = c.viterbi(punct, mask)
    c.close()
    for k in range(3):
        assert (got[k] == np.packbits(O.viterbi(punct[k]))).all()


@pytest.mark.parametrize("mode,fps", [(None, 1), (1, 16)])
def test_over_long_subchannel_inside_a_multiplex(mode, fps):
    """A 768 kbit/s sub-channel next to a small one and the FIC in one dabgpu_decode_frames call: one frame at a time
    (separate launches: the long one cannot join the grouped wave launch) and as a whole-group batch through the
    grouped lane launch (its traceback writes the long entry's words directly, the others through the LDS tile)."""
    big, small = dabgpu.subchannel(0, 768, level=2), dabgpu.subchannel(768, 64, level=3)
    assert big.length == 768 and small.length == 48
    rng = np.random.default_rng(11)
    soft = rng.integers(-127, 128, size=(fps, dabgpu.NB_FRAME_BITS), dtype=np.int8)
    c = make_ctx(mode, 32)
    fib, ok, outs, _ = c.decode_frames(soft, 1, [big, small])
    ref_fib, ref_ok = c.fic_decode(soft)
    assert (fib == ref_fib).all() and (ok == ref_ok).all()
    c.close()
    w = make_ctx(0, 32)                                       # reference: one call per sub-channel
    for i, sc in enumerate((big, small)):
        one, _ = w.msc_decode(sc, soft, 1)
        assert (outs[i] == one).all(), i
    w.close()


def test_every_entry_point_accepts_empty_batches():
    """Zero frames / streams / codewords / super-frames: DABGPU_OK, nothing written, no launch."""
    import torch
    c = make_ctx(None, 8)
    L = dabgpu.lib()
    dev = torch.device("cuda", 0)
    buf = torch.zeros((1 << 20,), dtype=torch.uint8, device=dev)
    d, h = buf.data_ptr(), c._h
    sc = dabgpu.subchannel(0, 64, level=3)
    import ctypes as C
    mask = (C.c_uint8 * (4 * 102))(*([1] * 408))
    assert L.dabgpu_ofdm_demod_frames_dev(h, d, 196608, 0, None, d, None, None, None) == 0
    assert L.dabgpu_ofdm_demod_streams_dev(h, d, 196608, 0, 4, 0.5, d, None, None, None) == 0
    assert L.dabgpu_fft_symbols_dev(h, d, 196608, 0, None, d, None) == 0
    assert L.dabgpu_sync_prs_dev(h, d, 196608, 0, None, 10, d, None) == 0
    assert L.dabgpu_acquire_dev(h, d, 0, 0, 100000, None, 4, d, d, None) == 0
    assert L.dabgpu_ofdm_demod_acquired_dev(h, d, 0, 0, 4, d, d, None, None, None) == 0
    assert L.dabgpu_fic_decode_dev(h, d, 230400, 0, d, d, None) == 0
    assert L.dabgpu_msc_decode_dev(h, C.byref(sc), d, 230400, 0, 4, None, None, d, None) == 0
    assert L.dabgpu_msc_decode_dev(h, C.byref(sc), d, 230400, 2, 0, None, None, d, None) == 0
    assert L.dabgpu_msc_decode_multi_dev(h, None, 0, d, 230400, 1, 1, None, None, None, None) == -1   # no list at all
    arr = (dabgpu.Subchannel * 1)(sc)
    outs = (C.c_void_p * 1)(d)
    assert L.dabgpu_msc_decode_multi_dev(h, arr, 1, d, 230400, 0, 4, None, None, outs, None) == 0
    assert L.dabgpu_decode_frames_dev(h, d, 230400, 0, 4, d, d, arr, 1, None, None, outs, None) == 0
    assert L.dabgpu_decode_frames_dev(h, d, 230400, 1, 0, d, d, None, 0, None, None, None, None) == 0
    assert L.dabgpu_viterbi_dev(h, d, 0, mask, 102, d, None) == 0
    assert L.dabgpu_dabplus_superframes_dev(h, d, 960, 0, 64, d, d, None) == 0
    assert L.dabgpu_streams_reset(h, 0) == 0
    # the ABI-v4 entry points: tracking, the decision-directed front end
    assert L.dabgpu_ofdm_demod_frames_dd_dev(h, d, 196608, 0, None, d, d, None) == 0
    assert L.dabgpu_track_start_dev(h, d, d, 0, 4, 196608, 0, None) == 0
    assert L.dabgpu_track_start_dev(h, d, d, 0, 4, 196608, 1, None) == 0
    assert L.dabgpu_ofdm_demod_tracked_dev(h, d, 0, 0, 400000, 4, 196608, None, d, None, None, d, d, None) == 0
    cfg = dabgpu.track_cfg(auto_acquire=1)
    assert L.dabgpu_ofdm_demod_tracked_dev(h, d, 0, 0, 400000, 4, 196608, C.byref(cfg), d, None, None, d, d, None) == 0
    # ... and their argument checks: no frame table, no frame slots, more streams than states
    assert L.dabgpu_ofdm_demod_tracked_dev(h, d, 0, 0, 400000, 4, 196608, None, d, None, None, None, d, None) == -1
    assert L.dabgpu_ofdm_demod_tracked_dev(h, d, 0, 0, 400000, 0, 196608, None, d, None, None, d, d, None) == -1
    assert L.dabgpu_ofdm_demod_tracked_dev(h, d, 400000, 3, 400000, 4, 196608, None, d, None, None, d, d, None) == -6
    assert L.dabgpu_track_start_dev(h, d, d, 3, 4, 196608, 0, None) == -6
    assert L.dabgpu_ofdm_demod_frames_dd_dev(h, d, 196608, 1, None, d, None, None) == -1      # no place for the sums
    c.sync()
    assert int(buf.sum().item()) == 0
    # the host-pointer variants
    z = np.zeros((0, dabgpu.NB_FRAME_BITS), np.int8)
    fib, ok = c.fic_decode(z)
    assert fib.shape[0] == 0 and ok.shape[0] == 0
    fib, ok, outs2, _ = c.decode_frames(z, 1, [sc])
    assert fib.shape[0] == 0
    c.close()
