#!/usr/bin/env python3
"""bench.py -- DAB Mode-I frames/s (OFDM front end + FIC Viterbi + one MSC subchannel): the timed step and the JSON line.

One step = one pass of the hot path over one batch of synthetic input that is already resident in HBM: `--ensembles`
independent DAB ensembles per GPU x `--frames` consecutive transmission frames each (default 64 x 256 = 16384 frames,
25.8 GB of cf32 IQ: BASELINE config 4, 64 ensembles per GPU, >= 256 frames per stream).  Per step:
  dabgpu_ofdm_demod_streams_dev -> dabgpu_decode_frames_dev (FIC + the sub-channel)
all through the C ABI (include/dabgpu.h) on the current torch stream.  torch is plumbing: device buffers, stream, events,
and torch.distributed (RCCL) for the barrier / max-reduce.

Nothing on the GPU side is told the frequency offsets the synthetic channel applied: the front end runs closed loop.  Per
ensemble the carrier offset is a whole number of carriers (|k| <= 3) plus a fraction (|f| <= 0.4); before the timed region
the coarse part is found on the first frame's phase reference symbol (dabgpu_sync_prs_dev) and the fine loop settles over
four untimed calls; during the timed steps every call corrects with the stream's state in HBM and updates it from the 76
cyclic-prefix correlations per frame the demodulation launch leaves (a small kernel after it, inside the timed region):
THE REFERENCE'S DATA FLOW AND ESTIMATOR (fine_freq_update_beta, /root/reference/src/render_radio_block.cpp:216) -- the
library's defaults.  `value` and `roofline` are that step.  The same step on the library's own decision-directed estimator
(opt-in; 17 % fewer bytes: of a frame's prefixes only the PRS's is read) is the `with_decision_directed_loop` leg, copied to
the top level as `value_own_estimator` / `roofline.frac_own_estimator`.

The IQ / soft-bit buffers are what the library gives any caller: ONE dabgpu_alloc_frame_buffers call, no policy here
(`config.buffer_placement` = its report).  `decoder.roofline` = the channel decoder against its own bounds (forward pass:
VALU issue; traceback: HBM), from the library's HIP events between its kernels during the timed steps.

Everything measured AFTER the timed region -- that leg, the sustained leg, the FFT stage, BASELINE configs 2 and 3 (one
ensemble), the host-fed figures, unaligned captures, the CPU baseline -- lives in bench_legs.py and runs on rank 0 of a
one-GPU run (`--legs all` forces it elsewhere, `--legs none` skips it).

Ensembles shard across GPUs with no data-path collective (weak scaling: 64 per rank; global ensemble ids
`id % world == rank`).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd"))
sys.path.insert(0, ROOT)

# algorithmic HBM bytes per frame (DESIGN.md "Measurement").  The timed step is the reference's data flow: all 76 symbol
# periods of a frame (prefixes included: the fine-frequency loop runs on the cyclic-prefix correlations) in, 230400 int8
# soft bits out.  SURVEY 8(d)'s A_ofdm = 1 803 264 B also counts the null symbol, which no launch reads.
A_OFDM = 76 * 2552 * 8 + 230400            # 1 782 016 B
A_OFDM_SURVEY = 196608 * 8 + 230400        # 1 803 264 B
HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: 8 TB/s spec
REALTIME_FPS = 1.0 / 0.096
ACS_FIC = 4 * 774 * 64                    # add-compare-selects per frame, FIC
ACS_MSC64 = 4 * 1542 * 64                 # one 64 kbit/s EEP 3-A subchannel (A12)


def make_streams(torch, dev, ids, n_frames, n_unique, snr_db, iq):
    """Fills `iq` ([len(ids) * n_frames][196608] cf32 on the device) with the ensembles `ids` (global ids); returns the
    carrier offsets the channel applied (kept on the host, for the CPU baseline only) and the transmitted multiplexes."""
    from dabgpu import synth
    ens = [synth.Ensemble(seed=0xDAB00000 + u, n_frames=4) for u in range(n_unique)]
    base = torch.from_numpy(np.stack([e.iq() for e in ens])).to(dev)        # [U][4][196608]
    n = torch.arange(synth.NB_FRAME_SAMPLES * n_frames, device=dev, dtype=torch.float64)
    iq = iq.view(len(ids), n_frames, synth.NB_FRAME_SAMPLES)
    sigma = float(np.sqrt(0.5 * 10 ** (-snr_db / 10)))
    cfos = []
    for s, gid in enumerate(ids):
        g = torch.Generator(device=dev)
        g.manual_seed(0xDAB0 + gid)
        r = np.random.default_rng(0xC0F0 + gid)
        cfo = (int(r.integers(-3, 4)) + float(r.uniform(-0.4, 0.4))) / 2048.0     # whole carriers + a fraction
        cfos.append(cfo)
        clean = base[gid % n_unique][torch.arange(n_frames, device=dev) % 4].reshape(-1)
        rot = torch.exp(2j * np.pi * cfo * n).to(torch.complex64)
        noise = torch.randn(clean.shape, generator=g, device=dev, dtype=torch.float32) + \
            1j * torch.randn(clean.shape, generator=g, device=dev, dtype=torch.float32)
        iq[s] = (clean * rot + sigma * noise).reshape(n_frames, -1)
    return np.asarray(cfos), [ens[gid % n_unique] for gid in ids]


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves.  This process has
    not imported torch nor made any HIP call, and it never will: it starts `python -m torch.distributed.run` as a
    fresh child (one rank per GPU, rendezvous on 127.0.0.1), hands rank 0's single JSON line on, and exits with the
    child's code.  A run that does not come back as exactly one line with n_gpus == N is an error, never a silent
    one-rank result."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: what RCCL needs on this driver
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    out, _ = p.communicate()
    lines = [l for l in out.splitlines() if l.startswith("{")]
    for l in out.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if p.returncode != 0:
        raise SystemExit(p.returncode)
    if len(lines) != 1:
        raise SystemExit("bench.py --gpus %d: expected one JSON line from rank 0, got %d" % (n, len(lines)))
    if json.loads(lines[0]).get("n_gpus") != n:
        raise SystemExit("bench.py --gpus %d: the job reports n_gpus = %r" % (n, json.loads(lines[0]).get("n_gpus")))
    print(lines[0])
    raise SystemExit(0)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--ensembles", type=int, default=64, help="independent ensembles per GPU")
    ap.add_argument("--frames", type=int, default=256, help="consecutive frames per ensemble per step (multiple of 4)")
    ap.add_argument("--unique", type=int, default=8, help="distinct synthetic multiplexes generated on the host")
    ap.add_argument("--snr", type=float, default=20.0)
    ap.add_argument("--legs", choices=["auto", "all", "none"], default="auto",
                    help="the measurements after the timed region (bench_legs.py): auto = all of them on a one-GPU run, none of "
                         "them when world > 1 (ranks 1..N-1 would idle in a barrier while rank 0 works for ~30 s)")
    ap.add_argument("--cpu-seconds", type=float, default=24.0, help="CPU baseline budget, both implementations together (0 = skip)")
    ap.add_argument("--sustained-seconds", type=float, default=3.0,
                    help="length of the `sustained` leg: the same step repeated for this long, so that the package's "
                         "power-limited steady state is in the record (0 = skip)")
    for leg, what in (("fft-stage", "the unfused FFT-stage measurement"), ("selective", "the selective-soft-output measurement"),
                      ("closed-loop", "the unaligned-capture / tracking measurement"), ("sustained", "the sustained leg"),
                      ("dd-leg", "the step on the library's own decision-directed estimator (with_decision_directed_loop / value_own_estimator)"),
                      ("single-ensemble", "BASELINE configs 2 and 3 (one ensemble) and the one-frame host path"),
                      ("host-fed", "the host-fed ring (64 frames per call from page-locked memory)"),
                      ("host-mirror", "the C++ host mirror end to end (dab_host_demo over three DAB+ services)"),
                      ("traffic", "measuring roofline.traffic now (two rocprofv3 --pmc child processes, ~20 s); the tracked figure "
                                  "of profiles/pmc_traffic.json is reported instead")):
        ap.add_argument("--no-" + leg, action="store_true", help="skip " + what)
    ap.add_argument("--decoder", choices=["auto", "lane", "wave"], default="auto",
                    help="which kernels decode (dabgpu_cfg.flags; results are identical): auto = by batch size, the library's default "
                         "(codeword-per-lane from 24 576 codewords per launch: the default shape); lane / wave pin one (tests: "
                         "the lane decoder's roofline record at a small shape)")
    ap.add_argument("--placement", choices=["plain", "domains"], default="domains",
                    help="what dabgpu_alloc_frame_buffers is asked for: two hipMallocs, or placement by HBM domain (<= 1.5 x the "
                         "pair held for ~0.1 s at set-up; the library itself ends in a plain pair on any failure and on a box "
                         "whose chunks share one domain -- config.buffer_placement is its report)")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    return args


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)                # never returns; nothing above this line has touched the GPU (or torch)

    import gc
    import torch
    import dabgpu
    from dabgpu import synth
    from dabgpu.shard import ensembles_of_rank, reduce_report, gather_per_rank
    # The cyclic collector stays off for the life of this short process: a generation-2 pass over the interpreter's few
    # hundred thousand objects takes ~20 ms, and one that lands at the head of a leg whose ten steps are 22 ms of GPU work
    # doubles that leg's wall clock (seen in `selective_soft_output`: 2.2 -> 3.9 ms per step in two runs of three).  Nothing
    # here builds reference cycles of any size.
    gc.collect()
    gc.disable()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d: launch with torch.distributed.run "
                         "--nproc-per-node == --gpus, or plainly as `python bench.py --gpus N`" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a gfx950 GPU (libdabgpu has no CPU fallback)")
    n_dev = torch.cuda.device_count()
    backend = os.environ.get("DABGPU_DIST_BACKEND", "nccl")      # "nccl" is RCCL on ROCm; "gloo" only for tests
    if world > n_dev and backend == "nccl":
        # RCCL refuses two ranks on one device; ranks sharing a GPU (tests on a one-GPU box) must say so explicitly
        raise SystemExit("bench.py --gpus %d: only %d device(s) visible (DABGPU_DIST_BACKEND=gloo lets test ranks share one)"
                         % (world, n_dev))
    dev_index = local_rank % n_dev        # (== local_rank on a real multi-GPU node; lets a 1-GPU box run 2 test ranks)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    # DABGPU_DIST_FORCE=1 (tests): take the process-group path even for one rank, so that the RCCL barrier and
    # reductions of the report execute on a single-GPU box
    if world > 1 or os.environ.get("DABGPU_DIST_FORCE") == "1":
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    red_dev = dev if backend == "nccl" else torch.device("cpu")
    dist_world = dist.get_world_size() if dist is not None else 1
    if dist_world != world:
        raise SystemExit("process group has %d ranks, WORLD_SIZE says %d" % (dist_world, world))
    run_legs = args.legs == "all" or (args.legs == "auto" and world == 1)

    E, F = args.ensembles, args.frames
    n_frames = E * F
    L = synth.NB_FRAME_SAMPLES
    ids = ensembles_of_rank(E * world, world, rank)              # this rank's share of the global ensemble list
    ctx = dabgpu.Context(device=dev_index, max_frames=n_frames,
                         flags={"auto": 0, "lane": dabgpu.FLAG_VITERBI_LANE, "wave": dabgpu.FLAG_VITERBI_WAVE}[args.decoder])
    torch.cuda.synchronize()
    tstream = torch.cuda.Stream(device=dev)      # non-null handle: the C ABI treats NULL as "context stream"
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    fib = torch.zeros((n_frames, 12, 32), dtype=torch.uint8, device=dev)
    crc = torch.zeros((n_frames, 12), dtype=torch.uint8, device=dev)
    sc = dabgpu.subchannel(0, 64, level=3)
    msc = torch.zeros((E, F * 4, 192), dtype=torch.uint8, device=dev)
    hist = [torch.zeros((E, 15, sc.length * 64), dtype=torch.int8, device=dev) for _ in range(2)]

    def decode_into(soft_buf, k=0):
        # FIC + the sub-channel of every frame: what BasicRadio::Process does, one call for the batch
        ctx.decode_frames_dev(soft_buf.data_ptr(), dabgpu.NB_FRAME_BITS, E, F, fib.data_ptr(), crc.data_ptr(), [sc],
                              [hist[k & 1].data_ptr()], [hist[(k & 1) ^ 1].data_ptr()], [msc.data_ptr()], stream)

    # ---- the IQ / soft-bit pair: ONE library call, no policy here (untimed set-up; config.buffer_placement is the
    # library's report verbatim, dabgpu.placement_report_dict).  Whether the pair is domain-aware or plain is the allocator's
    # decision (include/dabgpu.h, DABGPU_PLAIN_*): any caller of dabgpu_alloc_frame_buffers gets exactly these buffers.
    d_iq_base, d_soft_base, rep = ctx.alloc_frame_buffers(n_frames, L, dabgpu.PLACE_DOMAINS if args.placement == "domains"
                                                          else dabgpu.PLACE_PLAIN)
    placement = dabgpu.placement_report_dict(rep, requested=args.placement, final_bytes=n_frames * (L * 8 + dabgpu.NB_FRAME_BITS))
    iq = dabgpu.device_tensor(torch, d_iq_base, (n_frames, L), torch.complex64, dev)
    soft = dabgpu.device_tensor(torch, d_soft_base, (n_frames, dabgpu.NB_FRAME_BITS), torch.int8, dev)
    cfo_true, ens = make_streams(torch, dev, ids, F, min(args.unique, E * world), args.snr, iq)
    iq = iq.view(E, F, L)
    for h in hist:
        h.zero_()

    d_iq = iq.data_ptr() + synth.NB_NULL * 8          # first PRS sample of frame 0
    ofdm_ev, dec_ev = [], []
    BETA = 0.9                                        # fine_freq_update_beta, the reference's default order of magnitude

    # ---- the ceiling the front end is held against: a pure data mover of its own geometry on these very buffers (the
    # same runs, loads, stores and occupancy, no arithmetic; it leaves meaningless bytes in `soft`, which every later call
    # overwrites).  Untimed set-up.
    mover_ev = []
    for i in range(2 + 5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ctx.mover_frames_dev(d_iq, L, n_frames, soft.data_ptr(), True, stream)      # with_prefixes: what the timed launch reads
        e1.record()
        if i >= 2:
            mover_ev.append((e0, e1))
    torch.cuda.synchronize()
    mover_ms = float(np.mean([a.elapsed_time(b) for a, b in mover_ev]))

    # ---- acquisition, untimed: whole-carrier offset of every stream from its first PRS, then the fine loop settles
    ctx.streams_reset(E)
    sync_out = torch.zeros((E, 4), dtype=torch.int32, device=dev)
    ctx.sync_prs_dev(d_iq, F * L, E, None, 200, sync_out.data_ptr(), stream)
    torch.cuda.synchronize()
    coarse_found = sync_out[:, 0].cpu().numpy()
    for s in range(E):
        ctx.set_stream_offsets(s, coarse=-float(coarse_found[s]) / 2048.0)

    def step(k, timed):
        if timed:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ev[0].record()
        # (no correlation buffer passed: the 76 correlations per frame the loop runs on stay in the library's scratch)
        ctx.ofdm_demod_streams_dev(d_iq, L, E, F, BETA, soft.data_ptr(), None, None, stream)
        if timed:
            ev[1].record()
        decode_into(soft, k)
        if timed:
            ev[2].record()
            ofdm_ev.append((ev[0], ev[1])); dec_ev.append((ev[1], ev[2]))

    # settle the fine-frequency loop (part of acquisition, untimed) on the estimator the timed steps run: the library's
    # default, the cyclic-prefix correlations, which pull in from +-half a carrier
    for k in range(4):
        ctx.ofdm_demod_streams_dev(d_iq, L, E, F, BETA, soft.data_ptr(), None, None, stream)
    torch.cuda.synchronize()
    net = np.array([ctx.get_stats(s).net_freq_offset for s in range(E)])
    loop_residual = float(np.abs(net + cfo_true).max() * 2048.0)            # carriers; reported, not used

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k, False)
    barrier()
    # roofline.achieved is priced on the dominant kernel's own launches: HIP events recorded by the library around the
    # fused front-end launch alone, on the launch stream, during the timed steps (dabgpu_set_timing / _mean_kernel_ms)
    ctx.set_timing(True)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k, True)
    barrier()
    elapsed = time.perf_counter() - t0

    # ---- correctness of what was just timed (outside the timed region) ----
    fib_h, crc_h, msc_h = fib.cpu().numpy(), crc.cpu().numpy(), msc.cpu().numpy()
    fic_ok = bool(crc_h.all())
    msc_ok = True
    for s in range(E):
        e = ens[s]
        for f in range(F):
            fic_ok &= bool((fib_h[s * F + f] == e.fibs[f % 4]).all())
        # logical frame finished by CIF t was transmitted from CIF t-15 (cyclic 16-CIF multiplex);
        # with warm history every entry is valid
        for t in range(0 if args.warmup + args.steps >= 2 else 15, F * 4):
            msc_ok &= bool((msc_h[s, t] == e.msc_bytes[(t - 15) % 16]).all())
    elapsed_rank = elapsed
    elapsed, frames_total, (fic_ok, msc_ok) = reduce_report(dist, red_dev, elapsed, n_frames * args.steps,
                                                            [fic_ok, msc_ok])

    ofdm_call_ms = float(np.mean([a.elapsed_time(b) for a, b in ofdm_ev]))     # front-end call: kernel + state update
    # every timed step's own device time (front-end call + decode call), so that the spread inside a run -- and, over the
    # driver's records, between boxes -- is in the line
    step_dev_ms = np.array([a.elapsed_time(c) for (a, _), (_, c) in zip(ofdm_ev, dec_ev)])
    dec_ms = float(np.mean([a.elapsed_time(b) for a, b in dec_ev]))
    ofdm_ms, ofdm_launches = ctx.mean_kernel_ms(0)                             # the fused kernel's launches alone
    try:                                                                       # the grouped lane decode's kernels, one by one
        dec_parts = (ctx.mean_kernel_ms(4)[0], ctx.mean_kernel_ms(5)[0], ctx.mean_kernel_ms(6)[0], ctx.mean_kernel_ms(4)[1])
    except dabgpu.DabGpuError:
        dec_parts = None                                                       # (a batch below the lane decoder's threshold)
    ctx.set_timing(False)
    achieved = A_OFDM * n_frames / (ofdm_ms * 1e-3) / 1e9
    # one row per rank, so that an imbalance between the GPUs of a node is visible in the line
    # (+ what the allocator gave this rank and the mover's time on it: a slow rank of an 8-GPU node explains itself)
    per_rank = gather_per_rank(dist, red_dev, [n_frames * args.steps / elapsed_rank, ofdm_ms, dec_ms, dev_index,
                                               achieved / HBM_PEAK_GBS, mover_ms, rep.method, rep.fallback_reason,
                                               rep.pair_over_same_domain, rep.n_domains])
    collective = "none (one rank)"
    if dist is not None:
        collective = "nccl (RCCL)" if backend == "nccl" else backend + " (test ranks sharing a GPU: NOT RCCL)"
        # the process group has done all it is for (one barrier pair, three scalar reductions, one all_gather): it goes
        # away BEFORE anything long runs on rank 0, so that no rank sits in a collective while another works
        dist.barrier()
        dist.destroy_process_group()

    if rank == 0:
        value = frames_total / elapsed
        # HBM bytes per launch cannot be counted inside this run (PMC counters need rocprofv3 around the process, in
        # passes of their own): the figure is the one the tracked PMC run of this same command measured, and the line
        # says so; null when that file does not describe this launch shape
        traffic, traffic_source = None, "not measured in this run"
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("frames_per_launch") == n_frames and tj.get("mode") == "with cyclic-prefix correlations":
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_source = "profiles/pmc_traffic.json (rocprofv3 --pmc, separate run of this command: " \
                                     "TCC_EA0_RDREQ/WRREQ-derived FETCH_SIZE x 2 + WRITE_SIZE, tools/pmc_traffic.sh); not measured in this run"
            except Exception:
                traffic = None
        out = {
            "metric": "DAB Mode-I frames/sec (OFDM+Viterbi)", "value": value, "unit": "frames/s",
            "n_gpus": world, "world": dist_world, "collective_backend": collective,
            "per_rank": [{"rank": i, "frames_per_s": r[0], "front_end_kernel_ms": r[1], "decoder_ms": r[2], "device": int(r[3]),
                          "roofline_frac": r[4], "mover_same_geometry_ms": r[5],
                          "mover_source": "dabgpu_mover_frames_dev on this rank's timed buffers, mean of 5 launches after 2 (set-up, untimed)",
                          "kernel_over_mover": r[1] / r[5] if r[5] > 0 else None,
                          "buffer_placement": {"method": "domain-aware pair" if int(r[6]) == 1 else "plain hipMalloc pair",
                                               "fallback_reason": None if int(r[6]) == 1 else dabgpu.PLAIN_REASONS.get(int(r[7]), str(int(r[7]))),
                                               "pair_over_same_domain": round(r[8], 3), "domains_seen": int(r[9])}}
                         for i, r in enumerate(per_rank)],
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "step_ms": {"min": float(step_dev_ms.min()), "median": float(np.median(step_dev_ms)), "max": float(step_dev_ms.max()),
                        "what": "device time of each timed step on rank 0 (HIP events around the two calls)"},
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%d ensembles/GPU x %d frames/step, Mode-I OFDM + FIC Viterbi + one 64 kbps "
                                   "EEP-3A MSC subchannel, IQ resident in HBM" % (E, F),
                       "ensembles_per_gpu": E, "frames_per_step_per_gpu": n_frames, "snr_db": args.snr,
                       "carrier_offset": "unknown to the receiver: k + f carriers per ensemble, |k| <= 3, |f| <= 0.4",
                       "frequency_correction": "closed loop on the device (dabgpu_ofdm_demod_streams_dev, library defaults): coarse from "
                                               "the first PRS, fine: the reference's estimator -- mean angle of the 76 cyclic-prefix "
                                               "correlations of every frame of the previous call, fine_freq_update_beta 0.9",
                       "estimator": "cyclic-prefix correlations (the reference's; the library's default).  The library's own "
                                    "decision-directed loop is opt-in and reported as value_own_estimator / with_decision_directed_loop",
                       "sharding": "independent ensembles per rank (global id % world == rank), no data-path collective",
                       "buffer_placement": placement},
            "x_realtime": value / REALTIME_FPS,
            "fic_bit_exact": fic_ok, "msc_bit_exact": msc_ok,
            "fine_loop_residual_carriers": loop_residual,
            "value_own_estimator": None,
            "roofline": {"bound": "hbm", "kernel": "dabk::ofdm_wave_kernel<false,false,false,true> (fused A2..A6, cyclic-prefix correlations out)",
                         "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source, "avg_launch_ms": ofdm_ms, "launches_timed": ofdm_launches,
                         "front_end_call_ms": ofdm_call_ms, "frames_per_launch": n_frames,
                         "algorithmic_bytes_per_frame": A_OFDM,
                         "algorithmic_bytes_note": "76 x 2552 cf32 in (every symbol period, prefixes included), 230400 int8 out; priced on "
                                                   "SURVEY 8(d)'s A_ofdm (1 803 264 B: + the null symbol, which is never read) the same "
                                                   "launch would read achieved_on_survey_bytes",
                         "achieved_on_survey_bytes": A_OFDM_SURVEY * n_frames / (ofdm_ms * 1e-3) / 1e9,
                         # the practical ceiling: dabgpu_mover_frames_dev on the timed buffers -- the kernel's loads, stores,
                         # runs and occupancy without its arithmetic
                         "mover_same_geometry_ms": mover_ms,
                         "mover_same_geometry_GBps": A_OFDM * n_frames / (mover_ms * 1e-3) / 1e9,
                         # this box's ceiling for this read / write mix, on the same scale as `frac`
                         "box_mover_frac": A_OFDM * n_frames / (mover_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "kernel_over_mover": ofdm_ms / mover_ms,
                         "frac_own_estimator": None},
            # the channel decoder is integer add-compare-select work, not bandwidth: report ACS/s
            "decoder": {"fic_and_msc_ms": dec_ms, "acs_per_s": (ACS_FIC + ACS_MSC64) * n_frames / (dec_ms * 1e-3),
                        "entry_point": "dabgpu_decode_frames_dev (FIC + sub-channel codewords in one grouped launch)",
                        "kernels": args.decoder,
                        # the lane decoder against its bounds: forward pass VALU issue, traceback HBM (bench_legs.decoder_roofline);
                        # null when the batch is decoded by the wave-per-codeword kernels (below 24 576 codewords per launch)
                        "roofline": None},
            "cpu_baseline": None,
        }
        import bench_legs
        if dec_parts is not None:
            dr = bench_legs.decoder_roofline(dec_parts, n_frames, [(4 * n_frames, 774), (4 * n_frames, 1542)])
            dpath = os.path.join(ROOT, "profiles", "pmc_traffic_decoder.json")          # the tracked figure, until a leg measures it now
            try:
                dj = json.load(open(dpath))
                if dj.get("frames_per_call") == n_frames:
                    bench_legs.decoder_traffic(dr, {k.replace("dabk::", ""): v for k, v in dj["kernels"].items()},
                                               "profiles/pmc_traffic_decoder.json (rocprofv3 --pmc, a separate run: tools/pmc_decoder.sh); "
                                               "not measured in this run")
            except Exception:
                pass
            out["decoder"]["roofline"] = dr
        if run_legs:
            B = SimpleNamespace(torch=torch, dabgpu=dabgpu, synth=synth, ctx=ctx, dev=dev, stream=stream, args=args, E=E, F=F, L=L,
                                n_frames=n_frames, iq=iq, soft=soft, fib=fib, crc=crc, msc=msc, hist=hist, sc=sc, ens=ens,
                                d_iq=d_iq, BETA=BETA, step=step, decode_into=decode_into, ofdm_ev=ofdm_ev, dec_ev=dec_ev, net=net,
                                cfo_true=cfo_true, fib_h=fib_h, crc_h=crc_h, msc_h=msc_h, ms_per_step=elapsed / args.steps * 1e3,
                                dev_index=dev_index)
            bench_legs.run(B, out)
        else:
            out["legs"] = "skipped (world > 1: `--legs all` runs them on rank 0 after the process group is gone)" if world > 1 \
                else "skipped (--legs none)"
        print(json.dumps(out))
    d_bufs = (iq.data_ptr(), soft.data_ptr())
    del iq, soft
    ctx.free_frame_buffers(*d_bufs)
    ctx.close()


if __name__ == "__main__":
    main()
