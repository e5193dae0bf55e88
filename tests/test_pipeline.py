"""The host-fed ring (dabgpu_pipe_*): batches submitted asynchronously give exactly what the synchronous host-pointer calls
give on the same samples -- soft bits, FIBs, CRC flags and sub-channel bytes, with the time de-interleaver continued from
batch to batch on the device -- whatever the number of slots, with explicit offsets and closed loop."""
import numpy as np
import pytest

import dabgpu
from dabgpu import synth
from conftest import make_ctx

pytestmark = pytest.mark.gpu

L = dabgpu.FRAME_USED_SAMPLES


def _stream(ensemble, ensemble_iq, n_frames, snr_db, cfo, seed):
    """n_frames consecutive frames (the 5-frame multiplex tiled) through a channel -> [n_frames][76*2552] from the PRS on"""
    reps = (n_frames + ensemble.n_frames - 1) // ensemble.n_frames
    tx = np.tile(ensemble_iq, (reps, 1))[:n_frames]
    rx = synth.channel(tx.ravel(), snr_db=snr_db, cfo=cfo, rng=np.random.default_rng(seed)).reshape(tx.shape)
    return np.ascontiguousarray(rx[:, synth.NB_NULL:synth.NB_NULL + L])


def _reference(c, frames, fo, n_streams, scs, batch):
    """the synchronous calls, batch by batch, de-interleaver history carried by the host"""
    n = frames.shape[0]
    softs, fibs, oks, outs = [], [], [], [[] for _ in scs]
    hist = [None] * len(scs)
    for lo in range(0, n, batch):
        soft, _, _ = c.ofdm_demod_frames(frames[lo:lo + batch], fo[lo:lo + batch])
        fib, ok, out, hist = c.decode_frames(soft, n_streams, scs, history_in=hist, want_history=True)
        softs.append(soft); fibs.append(fib); oks.append(ok)
        for k in range(len(scs)):
            outs[k].append(out[k])
    return np.concatenate(softs), np.concatenate(fibs), np.concatenate(oks), [np.concatenate(o, axis=1) for o in outs]


@pytest.mark.parametrize("slots", [2, 3])
def test_ring_equals_the_synchronous_calls(built, ensemble, ensemble_iq, slots):
    n, batch = 40, 8
    cfo = 0.12 / 2048
    frames = _stream(ensemble, ensemble_iq, n, 14.0, cfo, 21)
    fo = np.full(n, -cfo, np.float32)
    sc = dabgpu.subchannel(ensemble.start_cu, 64, level=3)
    c = make_ctx(None, 64)
    soft_ref, fib_ref, ok_ref, out_ref = _reference(c, frames, fo, 1, [sc], batch)
    assert ok_ref.all()
    # page-locked buffers on both sides, every batch its own (they stay untouched until its wait)
    p_iq = dabgpu.PinnedArray((n, L), np.complex64); p_iq.array[:] = frames
    p_fo = dabgpu.PinnedArray((n,), np.float32); p_fo.array[:] = fo
    p_soft = dabgpu.PinnedArray((n, dabgpu.NB_FRAME_BITS), np.int8)
    p_fib = dabgpu.PinnedArray((n, 12, 32), np.uint8)
    p_ok = dabgpu.PinnedArray((n, 12), np.uint8)
    p_out = dabgpu.PinnedArray((n * 4, 192), np.uint8)
    c.pipe_open(slots, batch, L)
    tickets = []
    for lo in range(0, n, batch):
        t = c.pipe_submit(p_iq.array[lo:lo + batch], 1, batch, p_fo.array[lo:lo + batch], [sc], p_soft.array[lo:lo + batch],
                          p_fib.array[lo:lo + batch], p_ok.array[lo:lo + batch], [p_out.array[4 * lo:4 * (lo + batch)]])
        tickets.append(t)
    assert tickets == list(range(n // batch))
    for t in reversed(tickets):                                  # any order; an old ticket whose slot was reused returns at once
        c.pipe_wait(t)
    assert (p_soft.array == soft_ref).all()
    assert (p_fib.array == fib_ref).all() and (p_ok.array == ok_ref).all()
    assert (p_out.array == out_ref[0][0]).all()
    # the transmitted bytes, where the de-interleaver has filled (15 CIFs after the start of the stream)
    for t in range(15, 4 * n):
        assert (p_out.array[t] == ensemble.msc_bytes[(t - 15) % (4 * ensemble.n_frames)]).all(), t
    # a reset drops the rings: the same batches again start from erasures again and give the same bytes
    c.pipe_reset()
    first = p_out.array[:4 * batch].copy()
    p_out.array[:4 * batch] = 0
    t = c.pipe_submit(p_iq.array[:batch], 1, batch, p_fo.array[:batch], [sc], None, p_fib.array[:batch], p_ok.array[:batch],
                      [p_out.array[:4 * batch]])
    c.pipe_wait(t)
    assert (p_out.array[:4 * batch] == first).all()
    # a sub-channel listed twice would share one ring (keyed by start and size): refused before any ring is touched, and
    # the stream continues -- the second batch after the refusal equals the reference's second batch
    with pytest.raises(dabgpu.DabGpuError) as e:
        c.pipe_submit(p_iq.array[:batch], 1, batch, p_fo.array[:batch], [sc, sc], None, p_fib.array[:batch], p_ok.array[:batch],
                      [p_out.array[:4 * batch], p_out.array[:4 * batch]])
    assert e.value.status == -1
    second = p_out.array[4 * batch:8 * batch].copy()
    p_out.array[4 * batch:8 * batch] = 0
    t = c.pipe_submit(p_iq.array[batch:2 * batch], 1, batch, p_fo.array[batch:2 * batch], [sc], None, p_fib.array[batch:2 * batch],
                      p_ok.array[batch:2 * batch], [p_out.array[4 * batch:8 * batch]])
    c.pipe_wait(t)
    assert (p_out.array[4 * batch:8 * batch] == second).all()
    with pytest.raises(dabgpu.DabGpuError):
        c.pipe_wait(t + 1)                                       # never issued
    with pytest.raises(dabgpu.DabGpuError):
        c.pipe_submit(p_iq.array[:batch + 1], 1, batch + 1, p_fo.array[:batch + 1], [], None, p_fib.array[:batch + 1],
                      p_ok.array[:batch + 1], [])                # more frames than the ring was opened for
    with pytest.raises(dabgpu.DabGpuError):
        c.pipe_open(2, batch, L)                                 # one ring per context
    c.pipe_close()
    c.pipe_open(2, 4, L)                                         # ... and a new one after the close
    c.pipe_close()
    for p in (p_iq, p_fo, p_soft, p_fib, p_ok, p_out):
        p.close()
    c.close()


@pytest.mark.parametrize("dd", [False, True])
def test_ring_closed_loop_and_pageable_buffers(built, ensemble, ensemble_iq, dd):
    """freq_offset == NULL: the ring runs the stream call's loop (on the cyclic-prefix correlations, or decision-directed)
    on the context's stream states (two streams here), batch after batch -- soft bits and states equal those of
    dabgpu_ofdm_demod_streams_dev on the same samples; plain numpy (pageable) buffers are accepted."""
    import torch
    dev = torch.device("cuda", 0)
    n_streams, fps, batches = 2, 4, 3
    cfos = [0.2 / 2048, -0.3 / 2048]
    per = [_stream(ensemble, ensemble_iq, fps * batches, 16.0, cfos[s], 30 + s) for s in range(n_streams)]
    sc = dabgpu.subchannel(ensemble.start_cu, 64, level=3)
    a, b = make_ctx(None, 64), make_ctx(None, 64)
    for c in (a, b):
        c.streams_reset(n_streams)
        c.set_stream_loop(decision_directed=dd)
    b.pipe_open(3, n_streams * fps, L)
    keep = []
    for k in range(batches):
        batch = np.ascontiguousarray(np.concatenate([per[s][k * fps:(k + 1) * fps] for s in range(n_streams)]))
        d_iq = torch.from_numpy(batch).to(dev)
        d_soft = torch.zeros((n_streams * fps, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev)
        torch.cuda.synchronize()
        a.ofdm_demod_streams_dev(d_iq.data_ptr(), L, n_streams, fps, 0.9, d_soft.data_ptr(), None, None)
        a.sync()
        soft_a = d_soft.cpu().numpy()
        fib = np.zeros((n_streams * fps, 12, 32), np.uint8)
        ok = np.zeros((n_streams * fps, 12), np.uint8)
        soft_b = np.zeros((n_streams * fps, dabgpu.NB_FRAME_BITS), np.int8)
        out = np.zeros((n_streams, fps * 4, 192), np.uint8)
        t = b.pipe_submit(batch, n_streams, fps, None, [sc], soft_b, fib, ok, [out])
        keep.append((batch, soft_a, soft_b, fib, ok, out, t))
    for batch, soft_a, soft_b, fib, ok, out, t in keep:
        b.pipe_wait(t)
        assert (soft_a == soft_b).all()
    for s in range(n_streams):
        sa, sb = a.get_stats(s), b.get_stats(s)
        assert sa.fine_freq_offset == sb.fine_freq_offset and sa.total_frames_read == sb.total_frames_read == fps * batches
        assert abs(sb.fine_freq_offset * 2048 + cfos[s] * 2048) < 0.02
    # the last batch decodes (the loop has pulled in by then)
    assert keep[-1][4].all()
    b.pipe_close()
    a.close(); b.close()
