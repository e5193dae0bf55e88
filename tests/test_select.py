"""Soft-bit selection: the front end writes only the parts of the frame a batch receiver will decode.  What is
written must be byte-identical to the whole-frame run, what is not selected must stay untouched, and the channel
decoder must not notice the difference."""
import numpy as np
import pytest
import torch

import dabgpu
from conftest import make_ctx
from dabgpu import synth

pytestmark = pytest.mark.gpu


def test_selection_helper_and_argument_checks(built):
    sc = dabgpu.subchannel(10, 64, level=3)                 # 48 CUs from CU 10
    sel = dabgpu.soft_selection([sc])
    assert sel[0] == (0, 9216)
    assert sel[1:] == [(9216 + c * 55296 + 640, 3072) for c in range(4)]
    assert dabgpu.soft_selection([], with_fic=False) == []
    c = make_ctx(None, 4)
    L = dabgpu.lib()
    for bad in ([(8, 16)], [(0, 24)], [(-16, 16)], [(230400 - 16, 32)], [(0, 230416)]):
        with pytest.raises(dabgpu.DabGpuError):
            c.set_soft_selection(bad)
    assert L.dabgpu_ofdm_set_soft_selection(c._h, None, 3) == -1
    c.set_soft_selection([(0, 230400)])
    c.set_soft_selection(None)
    c.close()


@pytest.mark.parametrize("parts", [0, 1, 7])
def test_selected_output_equals_whole_frame_run(built, ensemble, ensemble_iq, parts):
    c = make_ctx(None, 8, ofdm_symbol_runs=parts)
    rng = np.random.default_rng(3)
    rx = synth.channel(ensemble_iq.ravel(), snr_db=15.0, cfo=0.21 / 2048, rng=rng).reshape(ensemble_iq.shape)
    frames = np.ascontiguousarray(rx[:, synth.NB_NULL:])
    n = frames.shape[0]
    fo = np.full(n, -0.21 / 2048, np.float32)
    full, _, _ = c.ofdm_demod_frames(frames, fo)
    sc = dabgpu.subchannel(ensemble.start_cu, 64, level=3)
    # FIC + the sub-channel + an odd little piece in the middle of a symbol
    sel = dabgpu.soft_selection([sc]) + [(100000 - 100000 % 16, 48)]
    c.set_soft_selection(sel)
    dev = torch.device("cuda", 0)
    d_iq = torch.from_numpy(frames).to(dev)
    d_fo = torch.from_numpy(fo).to(dev)
    d_soft = torch.full((n, dabgpu.NB_FRAME_BITS), 99, dtype=torch.int8, device=dev)
    c.ofdm_demod_frames_dev(d_iq.data_ptr(), frames.shape[1], n, d_fo.data_ptr(), d_soft.data_ptr(), None, None, None)
    c.sync()
    got = d_soft.cpu().numpy()
    want = np.full_like(full, 99)
    for first, count in sel:
        want[:, first:first + count] = full[:, first:first + count]
    assert (got == want).all()
    # the decision-directed sums under a selection: the symbols whose bits are wanted contribute, nothing else
    from oracle import oracle as O
    wanted = sorted({(b // 3072) + 1 for first, count in sel for b in range(first, first + count, 16)})
    d_soft2 = torch.full((n, dabgpu.NB_FRAME_BITS), 99, dtype=torch.int8, device=dev)
    d_dd = torch.zeros((n, 76), dtype=torch.complex64, device=dev)
    torch.cuda.synchronize()
    c.ofdm_demod_frames_dd_dev(d_iq.data_ptr(), frames.shape[1], n, d_fo.data_ptr(), d_soft2.data_ptr(), d_dd.data_ptr())
    c.sync()
    assert (d_soft2.cpu().numpy() == want).all()
    dd = d_dd.cpu().numpy()
    for f in range(2):
        ref = O.ofdm_demod_frame_dd(frames[f], float(fo[f]))[1][wanted].astype(np.complex128).sum()
        assert abs(dd[f, 1:].astype(np.complex128).sum() - ref) <= 1e-4 * abs(ref), (f, parts)
        assert abs(dd[f, 0] - O.ofdm_demod_frame_dd(frames[f], float(fo[f]))[1][0]) <= 1e-3 * abs(dd[f, 0])
    # ... and with a selection that leaves the PRS untransformed (no FIC, nothing in symbol 1) its prefix correlation
    # still arrives in entry 0 (it then comes from the prefix and the symbol's last 512 samples alone)
    c.set_soft_selection([(100000 - 100000 % 16, 48)])
    d_dd.zero_()
    torch.cuda.synchronize()
    c.ofdm_demod_frames_dd_dev(d_iq.data_ptr(), frames.shape[1], n, d_fo.data_ptr(), d_soft2.data_ptr(), d_dd.data_ptr())
    c.sync()
    dd = d_dd.cpu().numpy()
    for f in range(2):
        odd = O.ofdm_demod_frame_dd(frames[f], float(fo[f]))[1]
        assert abs(dd[f, 0] - odd[0]) <= 1e-3 * abs(odd[0])
        ref = odd[100000 // 3072 + 1].astype(np.complex128)
        assert abs(dd[f, 1:].astype(np.complex128).sum() - ref) <= 1e-4 * abs(ref)
    c.set_soft_selection(sel)
    # the host-pointer call copies back the selected runs only: the rest of the caller's buffer stays as it was
    host = np.full_like(full, 99)
    c.ofdm_demod_frames(frames, fo, soft=host)
    assert (host == want).all()
    # the decoder sees no difference
    fib_a, ok_a = c.fic_decode(full)
    fib_b, ok_b = c.fic_decode(got)
    assert (fib_a == fib_b).all() and (ok_a == ok_b).all() and ok_a.all()
    out_a, _ = c.msc_decode(sc, full, n_streams=1, want_history=True)
    out_b, _ = c.msc_decode(sc, got, n_streams=1, want_history=True)
    assert (out_a == out_b).all()
    # the constellation output needs every symbol: the selection is ignored there
    soft2, _, dq = c.ofdm_demod_frames(frames[:1], fo[:1], want_dqpsk=True)
    assert (soft2 == full[:1]).all() and np.isfinite(dq).all()
    # and switching it off restores whole frames
    c.set_soft_selection(None)
    again, _, _ = c.ofdm_demod_frames(frames, fo)
    assert (again == full).all()
    c.close()
