#!/usr/bin/env python3
"""bench.py -- DAB Mode-I frames/s (OFDM front end + FIC Viterbi + one MSC subchannel).

One step = one pass of the hot path over one batch of synthetic input that is already
resident in HBM: `--ensembles` independent DAB ensembles per GPU x `--frames` consecutive
transmission frames each (default 64 x 256 = 16384 frames, 25.8 GB of cf32 IQ: BASELINE config 4,
64 ensembles per GPU, >= 256 frames per stream).  Per step:
  dabgpu_ofdm_demod_streams_dev -> dabgpu_decode_frames_dev (FIC + the sub-channel)
all through the C ABI (include/dabgpu.h) on the current torch stream.  torch is plumbing:
device buffers, stream, events, and torch.distributed (RCCL) for the barrier / max-reduce.

Nothing on the GPU side is told the frequency offsets the synthetic channel applied: the front end runs closed
loop.  Per ensemble the carrier offset is a whole number of carriers (|k| <= 3) plus a fraction (|f| <= 0.4); before
the timed region the coarse part is found on the first frame's phase reference symbol (dabgpu_sync_prs_dev) and the
fine loop settles over four untimed calls; during the timed steps every call corrects with the stream's state in
HBM and updates it from the decision-directed sums the demodulation launch leaves (a small kernel after it, inside the
timed region; `with_cyclic_prefix_correlations` is the same step on the reference's estimator).  `closed_loop` repeats the step on the same samples presented as unaligned captures: null-symbol
search, per-frame frequency and timing from the PRS, demodulation where the frames lie.

Ensembles shard across GPUs with no data-path collective (weak scaling: 64 per rank; global ensemble ids
`id % world == rank`).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd"))
sys.path.insert(0, ROOT)

# algorithmic HBM bytes per frame (DESIGN.md "Measurement").  Round 3: the closed-loop front end no longer reads the
# cyclic prefixes of the data symbols (decision-directed fine-frequency loop): the useful 2048 samples of the 76 symbols
# and the prefix of the PRS in, 230400 int8 soft bits out.  A_OFDM_CP is what the kernel moves when the caller asks for the cyclic-prefix correlations
# (rounds 1-2, and the `with_cyclic_prefix_correlations` leg below); SURVEY 8(d)'s A_ofdm = 1 803 264 B also counts the
# null symbol, which no kernel here ever read.
A_OFDM = (76 * 2048 + 504) * 8 + 230400    # 1 479 616 B (the PRS keeps its prefix: it resolves the estimator's ambiguity)
A_OFDM_CP = 76 * 2552 * 8 + 230400         # 1 782 016 B
A_OFDM_SURVEY = 196608 * 8 + 230400        # 1 803 264 B
A_FFT = 76 * 2552 * 8 + 76 * 2048 * 8      # 2 796 800 B (unfused FFT stage, SURVEY 8(d): prefixes counted)
A_FFT_MOVED = 2 * 76 * 2048 * 8            # 2 490 368 B (what the FFT-stage kernel reads and writes)
HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: 8 TB/s spec
REALTIME_FPS = 1.0 / 0.096
ACS_FIC = 4 * 774 * 64                    # add-compare-selects per frame, FIC
ACS_MSC64 = 4 * 1542 * 64                 # one 64 kbit/s EEP 3-A subchannel (A12)


def place_buffers(torch, dabgpu, ctx, dev, n_frames, mode, n_candidates):
    """Where the IQ and soft-bit buffers of this rank live.  MI355X's HBM behaves as three domains of 96 GB (large
    contiguous address ranges; profiles/r02_hbm_domains.txt maps them): a launch that reads from and writes to the SAME
    domain pays ~12 % for the read/write turn-arounds (5.7 instead of 5.0 ms for a data mover of this kernel's shape),
    while reading alone or writing alone runs at the same rate everywhere.  Which domain an allocation lands in is the
    driver's choice, so the library places the pair itself:
      "placed"      (default) dabgpu_alloc_frame_buffers_placed: physical memory in 1 GiB chunks through the virtual
                    memory API, each chunk's domain found with a small data mover (~30 ms), IQ mapped over one domain and
                    the soft bits over another; never more than 1.2 x the final footprint held
      "candidates"  round 2's dabgpu_alloc_frame_buffers: n candidates of each buffer (4 x 29.6 GB), the front end timed
                    on every pair, the fastest kept
      "plain"       two hipMallocs
    Untimed set-up, reported in `config`."""
    L = dabgpu.NB_FRAME_SAMPLES
    report = None
    if mode == "placed":
        d_iq, d_soft, rep = ctx.alloc_frame_buffers_placed(n_frames, L)
        kept = (0, 0)
        report = {"method": "domain-aware arena" if rep.method == 1 else "plain allocation (buffers too small for placement, or "
                  "no virtual-memory API)", "chunks_taken": rep.n_chunks, "chunk_bytes": int(rep.chunk_bytes),
                  "chunk_domains": rep.domains.decode(), "iq_chunk_domains": rep.iq_map.decode(),
                  "soft_chunk_domains": rep.soft_map.decode(), "domains_seen": rep.n_domains,
                  "soft_bits_written_beside_same_domain_reads_per_mille": rep.conflicts, "classify_ms": round(rep.classify_ms, 2),
                  "mover_on_pair_over_mover_in_one_domain": round(float(rep.pair_over_same_domain), 3),
                  "setup_peak_bytes": int(rep.setup_peak_bytes),
                  "setup_peak_over_final_footprint": round(rep.setup_peak_bytes / (n_frames * (L * 8 + dabgpu.NB_FRAME_BITS)), 3),
                  "front_end_ms_on_placed_pair": round(rep.front_end_ms, 3)}
    else:
        d_iq, d_soft, table, kept = ctx.alloc_frame_buffers(n_frames, L, n_candidates if mode == "candidates" else 1)
        if table is not None:
            flat = [float(x) for r in table for x in r]
            report = {"method": "timed candidates", "candidates": n_candidates,
                      "probe_front_end_ms": [[round(float(x), 3) for x in r] for r in table], "kept": list(kept),
                      "front_end_ms_plain_alloc": round(float(table[0][0]), 3),          # the pair a plain allocation would have got
                      "front_end_ms_kept_pair": round(float(table[kept[0]][kept[1]]), 3),
                      "probe_min_ms": round(min(flat), 3), "probe_max_ms": round(max(flat), 3)}
    iq = dabgpu.device_tensor(torch, d_iq, (n_frames, L), torch.complex64, dev)
    soft = dabgpu.device_tensor(torch, d_soft, (n_frames, dabgpu.NB_FRAME_BITS), torch.int8, dev)
    return iq, soft, report, kept


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


def copy_ceiling(torch, dev):
    """Measured device-to-device copy rate (read + write bytes), GB/s: the practical HBM ceiling next to the 8 TB/s
    spec figure."""
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=dev)
    b = torch.empty(n, dtype=torch.uint8, device=dev)
    for _ in range(2):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * n * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9


def fftw_fft_stage(iq_host, seconds=3.0):
    """if the box happens to have FFTW3f (the reference's FFT library,
    /root/reference/CMakeLists.txt:55-64), time its 2048-point c2c transform on the FFT stage's work (76 per frame,
    one thread, FFTW_MEASURE).  Returns frames/s or None when the library is not installed."""
    import ctypes as C
    try:
        fw = C.CDLL("libfftw3f.so.3")
    except OSError:
        return None
    fw.fftwf_malloc.restype = C.c_void_p
    fw.fftwf_malloc.argtypes = [C.c_size_t]
    fw.fftwf_plan_many_dft.restype = C.c_void_p
    fw.fftwf_plan_many_dft.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int,
                                       C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_uint]
    fw.fftwf_execute.argtypes = [C.c_void_p]
    fw.fftwf_destroy_plan.argtypes = [C.c_void_p]
    fw.fftwf_free.argtypes = [C.c_void_p]
    n, howmany = C.c_int(2048), 76
    nbytes = howmany * 2552 * 8
    a, b = fw.fftwf_malloc(nbytes), fw.fftwf_malloc(howmany * 2048 * 8)
    C.memmove(a, iq_host[0].ctypes.data, nbytes)
    # symbol l: input at l*2552 + 504, output at l*2048
    plan = fw.fftwf_plan_many_dft(1, C.byref(n), howmany, C.c_void_p(a + 504 * 8), None, 1, 2552, C.c_void_p(b), None, 1, 2048,
                                  -1, 0)      # FFTW_FORWARD, FFTW_MEASURE
    C.memmove(a, iq_host[0].ctypes.data, nbytes)
    t0, k = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        fw.fftwf_execute(plan)
        k += 1
    el = time.perf_counter() - t0
    fw.fftwf_destroy_plan(plan); fw.fftwf_free(a); fw.fftwf_free(b)
    return k / el


def cpu_quota_cores():
    """CPU time the container may use, in cores (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited/unknown."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(p)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except Exception:
        return None


def cpu_rows(O, iq_host, fo_host, sc_len_bits, mask, nsteps, budget_s, threads, simd):
    """The rows SURVEY.md 8d / BASELINE.md 4.3 ask for, for one CPU implementation:
      single_core_value        one thread doing OFDM demod + FIC + 4 MSC logical frames per frame
      ofdm_only_1_thread       BASELINE config 1: the front end alone on one thread
      as_deployed_1_plus_1     one OFDM thread feeding one decoder thread through a 2-frame ring
                               (/root/reference/src/dab_module.cpp:92, src/radio_block.cpp:23-44)
      value                    every logical CPU the scheduler lists, one frame stream per thread; `effective_cores` =
                               value / single_core_value says how many cores' worth of time the box actually granted"""
    n = iq_host.shape[0]
    k1, t1 = O.bench_frames_timed(iq_host, fo_host, budget_s * 0.15, 1, mask, nsteps, sc_len_bits, simd=simd)
    ko, to = O.bench_ofdm_only_timed(iq_host, fo_host, budget_s * 0.15, simd=simd)
    kp, tp = O.bench_pipeline_timed(iq_host, fo_host, budget_s * 0.2, mask, nsteps, sc_len_bits, simd=simd)
    total, tn = O.bench_frames_timed(iq_host, fo_host, budget_s * 0.5, threads, mask, nsteps, sc_len_bits, simd=simd)
    single = k1 / t1
    what = "oracle/simd_port.c" if simd else "oracle/dab_oracle.c"
    return {"value": total / tn, "unit": "frames/s", "cores": threads,
            "effective_cores": (total / tn) / single,
            "sample": "%d frames (OFDM+FIC+64kbps EEP-3A MSC, %d distinct bench-input frames cycled) through "
                      "%s on %d pthreads in %.1f s" % (total, n, what, threads, tn),
            "single_core_value": single,
            "ofdm_only_1_thread": {"value": ko / to, "unit": "frames/s", "sample": "%d frames in %.1f s (BASELINE config 1)" % (ko, to)},
            "as_deployed_1_plus_1": {"value": kp / tp, "unit": "frames/s", "threads": 2,
                                     "sample": "%d frames in %.1f s: one OFDM pthread -> 2-frame ring -> one decoder pthread" % (kp, tp)}}


def cpu_baseline(iq_host, fo_host, sc_len_bits, mask, nsteps, budget_s, threads, truth_fibs=None):
    """Two CPU implementations timed on the same bounded sample of the workload, on the host cores of the GPU box:
      "port"       the oracle (oracle/dab_oracle.c): scalar, libm sin/cos per sample, radix-2 FFT, exact int32 Viterbi --
                   the checker, timed as it is;
      "simd_port"  oracle/simd_port.c, the path written as a CPU implementation of the reference's class is written
                   (table-driven NCO, four-step FFT in AVX loops, 16-bit saturating AVX2 Viterbi; `-O3 -march=native
                   -ffast-math`, the reference's flags, built on this box): what "the reference FFTW3f/AVX2 path" would
                   be in the neighbourhood of.  FFTW3f itself is timed too when the box has it.
    Stated baselines, never the target."""
    from oracle import oracle as O
    fftw = fftw_fft_stage(iq_host)
    out = cpu_rows(O, iq_host, fo_host, sc_len_bits, mask, nsteps, budget_s * 0.5, threads, False)
    out.update({"kind": "port", "cgroup_cpu_quota_cores": cpu_quota_cores(),
                "fftw3f_fft_stage_frames_per_s_1_thread": fftw if fftw is not None else "FFTW3f: not available on this box"})
    try:
        simd = cpu_rows(O, iq_host, fo_host, sc_len_bits, mask, nsteps, budget_s * 0.5, threads, True)
        simd["kind"] = "simd_port"
        simd["isa"] = O.simd_isa()
        if truth_fibs is not None:                       # it must decode the bench's own inputs to the transmitted FIBs
            ok = True
            for f in range(min(4, iq_host.shape[0])):
                fib, crc = O.simd_fic_decode(O.simd_ofdm_demod_frame(iq_host[f], float(fo_host[f])))
                ok &= bool(crc.all()) and bool((fib == truth_fibs[f]).all())
            simd["decodes_bench_inputs_to_transmitted_fibs"] = ok
        out["simd_port"] = simd
    except Exception as e:                               # no compiler on the box: say so instead of dropping the row silently
        out["simd_port"] = "not available: %s" % e
    return out


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves.  This process has
    not imported torch.cuda nor made any HIP call, and it never will: it starts `python -m torch.distributed.run` as a
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--ensembles", type=int, default=64, help="independent ensembles per GPU")
    ap.add_argument("--frames", type=int, default=256, help="consecutive frames per ensemble per step (multiple of 4)")
    ap.add_argument("--unique", type=int, default=8, help="distinct synthetic multiplexes generated on the host")
    ap.add_argument("--snr", type=float, default=20.0)
    ap.add_argument("--cpu-seconds", type=float, default=24.0, help="CPU baseline budget, both implementations together (0 = skip)")
    ap.add_argument("--no-fft-stage", action="store_true", help="skip the unfused FFT-stage measurement")
    ap.add_argument("--no-selective", action="store_true", help="skip the extra selective-soft-output measurement")
    ap.add_argument("--no-closed-loop", action="store_true", help="skip the unaligned-capture closed-loop measurement")
    ap.add_argument("--sustained-seconds", type=float, default=3.0,
                    help="length of the extra `sustained` leg: the same step repeated for this long, so that the package's "
                         "power-limited steady state is in the record (0 = skip)")
    ap.add_argument("--no-sustained", action="store_true", help="skip the sustained leg")
    ap.add_argument("--no-cp-leg", action="store_true",
                    help="skip the leg that runs the step with the cyclic-prefix correlations (rounds 1-2's data flow)")
    ap.add_argument("--no-plain-compare", action="store_true",
                    help="skip timing the front end on a plainly allocated copy of the buffers beside the placed pair")
    ap.add_argument("--placement", choices=["placed", "candidates", "plain"], default="placed",
                    help="how the IQ / soft-bit buffers are placed in HBM (place_buffers)")
    ap.add_argument("--placement-candidates", type=int, default=4,
                    help="--placement candidates: buffers of each kind timed at set-up (four pairs of the default shape "
                         "span 118 GB, more than one 96 GB HBM domain); 1 = --placement plain")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)                # never returns; nothing above this line has touched the GPU

    import torch
    import dabgpu
    from dabgpu import synth
    from dabgpu.shard import ensembles_of_rank, reduce_report, gather_per_rank

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

    E, F = args.ensembles, args.frames
    n_frames = E * F
    ids = ensembles_of_rank(E * world, world, rank)              # this rank's share of the global ensemble list
    ctx = dabgpu.Context(device=dev_index, max_frames=n_frames)
    torch.cuda.synchronize()
    tstream = torch.cuda.Stream(device=dev)      # non-null handle: the C ABI treats NULL as "context stream"
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    fib = torch.zeros((n_frames, 12, 32), dtype=torch.uint8, device=dev)
    crc = torch.zeros((n_frames, 12), dtype=torch.uint8, device=dev)
    sc = dabgpu.subchannel(0, 64, level=3)
    msc = torch.zeros((E, F * 4, 192), dtype=torch.uint8, device=dev)
    hist = [torch.zeros((E, 15, sc.length * 64), dtype=torch.int8, device=dev) for _ in range(2)]
    cyc = torch.zeros((n_frames, 76), dtype=torch.complex64, device=dev)

    def decode_into(soft_buf, k=0):
        # FIC + the sub-channel of every frame: what BasicRadio::Process does, one call for the batch
        ctx.decode_frames_dev(soft_buf.data_ptr(), dabgpu.NB_FRAME_BITS, E, F, fib.data_ptr(), crc.data_ptr(), [sc],
                              [hist[k & 1].data_ptr()], [hist[(k & 1) ^ 1].data_ptr()], [msc.data_ptr()], stream)

    if args.placement == "candidates" and args.placement_candidates <= 1:
        args.placement = "plain"
    iq, soft, placement, kept_pair = place_buffers(torch, dabgpu, ctx, dev, n_frames, args.placement,
                                                   max(1, min(8, args.placement_candidates)))
    cfo_true, ens = make_streams(torch, dev, ids, F, min(args.unique, E * world), args.snr, iq)
    iq = iq.view(E, F, synth.NB_FRAME_SAMPLES)
    for h in hist:
        h.zero_()

    d_iq = iq.data_ptr() + synth.NB_NULL * 8          # first PRS sample of frame 0
    ofdm_ev, dec_ev = [], []
    BETA = 0.9                                        # fine_freq_update_beta, the reference's default order of magnitude

    # ---- acquisition, untimed: whole-carrier offset of every stream from its first PRS, then the fine loop settles
    ctx.streams_reset(E)
    sync_out = torch.zeros((E, 4), dtype=torch.int32, device=dev)
    ctx.sync_prs_dev(d_iq, F * synth.NB_FRAME_SAMPLES, E, None, 200, sync_out.data_ptr(), stream)
    torch.cuda.synchronize()
    coarse_found = sync_out[:, 0].cpu().numpy()
    for s in range(E):
        ctx.set_stream_offsets(s, coarse=-float(coarse_found[s]) / 2048.0)

    def step(k, timed):
        if timed:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ev[0].record()
        # (no correlation buffer passed: the fine loop runs decision-directed; of a frame's prefixes only the PRS's is read)
        ctx.ofdm_demod_streams_dev(d_iq, synth.NB_FRAME_SAMPLES, E, F, BETA, soft.data_ptr(), None, None, stream)
        if timed:
            ev[1].record()
        decode_into(soft, k)
        if timed:
            ev[2].record()
            ofdm_ev.append((ev[0], ev[1])); dec_ev.append((ev[1], ev[2]))

    # settle the fine-frequency loop (part of acquisition, untimed), decision-directed from the first call as the timed
    # steps run: the fourth-power estimate is exact to 1e-4 carriers but repeats every 0.2, the cyclic prefix of each
    # frame's PRS (the one prefix that is read) picks its branch, so the loop pulls in from +-half a carrier
    ctx.set_stream_loop(decision_directed=True)      # calls without a correlation buffer skip the other 75 prefixes
    for k in range(4):
        ctx.ofdm_demod_streams_dev(d_iq, synth.NB_FRAME_SAMPLES, E, F, BETA, soft.data_ptr(), None, None, stream)
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
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
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
    dec_ms = float(np.mean([a.elapsed_time(b) for a, b in dec_ev]))
    ofdm_ms, ofdm_launches = ctx.mean_kernel_ms(0)                             # the fused kernel's launches alone
    ctx.set_timing(False)
    # one row per rank, so that an imbalance between the GPUs of a node is visible in the line
    per_rank = gather_per_rank(dist, red_dev, [n_frames * args.steps / elapsed_rank, ofdm_ms, dec_ms, dev_index,
                                               kept_pair[0], kept_pair[1]])

    if rank == 0:
        value = frames_total / elapsed
        achieved = A_OFDM * n_frames / (ofdm_ms * 1e-3) / 1e9
        # HBM bytes per launch cannot be counted inside this run (PMC counters need rocprofv3 around the process, in
        # passes of their own): the figure is the one the tracked PMC run of this same command measured, and the line
        # says so; null when that file does not describe this launch shape
        traffic, traffic_source = None, "not measured in this run"
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("frames_per_launch") == n_frames and tj.get("mode") == "decision-directed, no cyclic prefix read":
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_source = "profiles/pmc_traffic.json (rocprofv3 --pmc, separate run of this command: " \
                                     "TCC_EA0_RDREQ/WRREQ-derived FETCH_SIZE x 2 + WRITE_SIZE, tools/pmc_traffic.sh); not measured in this run"
            except Exception:
                traffic = None
        out = {
            "metric": "DAB Mode-I frames/sec (OFDM+Viterbi)", "value": value, "unit": "frames/s",
            "n_gpus": world, "rccl_world": dist_world,
            "collective_backend": (backend + (" (RCCL over xGMI)" if backend == "nccl" else " (test ranks sharing a GPU)"))
                                  if dist is not None else "none (one rank)",
            "per_rank": [{"rank": i, "frames_per_s": r[0], "front_end_kernel_ms": r[1], "decoder_ms": r[2], "device": int(r[3]),
                          "kept_placement_pair": [int(r[4]), int(r[5])]} for i, r in enumerate(per_rank)],
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%d ensembles/GPU x %d frames/step, Mode-I OFDM + FIC Viterbi + one 64 kbps "
                                   "EEP-3A MSC subchannel, IQ resident in HBM" % (E, F),
                       "ensembles_per_gpu": E, "frames_per_step_per_gpu": n_frames, "snr_db": args.snr,
                       "carrier_offset": "unknown to the receiver: k + f carriers per ensemble, |k| <= 3, |f| <= 0.4",
                       "frequency_correction": "closed loop on the device (dabgpu_ofdm_demod_streams_dev): coarse from the first PRS, "
                                               "fine: decision-directed from the first (untimed) call on -- fourth power of the "
                                               "differential symbols of the previous call, its 0.2-carrier ambiguity resolved by the "
                                               "cyclic prefix of each frame's PRS, the only prefix that is read",
                       "sharding": "independent ensembles per rank (global id % world == rank), no data-path collective",
                       "buffer_placement": placement if placement is not None else "first allocation taken"},
            "x_realtime": value / REALTIME_FPS,
            "fic_bit_exact": fic_ok, "msc_bit_exact": msc_ok,
            "fine_loop_residual_carriers": loop_residual,
            "roofline": {"bound": "hbm", "kernel": "dabk::ofdm_wave_kernel<false,false,false,true> (fused A2..A6)", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source, "avg_launch_ms": ofdm_ms, "launches_timed": ofdm_launches,
                         "front_end_call_ms": ofdm_call_ms, "frames_per_launch": n_frames,
                         "algorithmic_bytes_per_frame": A_OFDM,
                         "algorithmic_bytes_note": "76 x 2048 cf32 + the PRS's 504-sample prefix in, 230400 int8 out: the other 75 cyclic "
                                                   "prefixes (19.5 % of the samples) are not read any more; priced on SURVEY 8(d)'s A_ofdm (1 803 264 B, prefixes "
                                                   "and null symbol included) the same launch would read achieved_on_survey_bytes",
                         "achieved_on_survey_bytes": A_OFDM_SURVEY * n_frames / (ofdm_ms * 1e-3) / 1e9,
                         "copy_ceiling": copy_ceiling(torch, dev)},
            # the channel decoder is integer add-compare-select work, not bandwidth: report ACS/s
            "decoder": {"fic_and_msc_ms": dec_ms, "acs_per_s": (ACS_FIC + ACS_MSC64) * n_frames / (dec_ms * 1e-3),
                        "entry_point": "dabgpu_decode_frames_dev (FIC + sub-channel codewords in one grouped launch)"},
        }
        if args.placement == "placed" and placement is not None and not args.no_plain_compare:
            # what two plain allocations would have given in this very process: the same front-end call (open loop, the
            # offsets the closed loop arrived at) on a torch-allocated pair holding the same samples, beside the placed pair
            fo_cmp = torch.from_numpy(np.repeat(net.astype(np.float32), F)).to(dev)
            iq_p = torch.empty((n_frames, synth.NB_FRAME_SAMPLES), dtype=torch.complex64, device=dev)
            soft_p = torch.empty((n_frames, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev)
            iq_p.copy_(iq.reshape(n_frames, -1))
            pairs = (("placed", iq.data_ptr(), soft.data_ptr()), ("plain_alloc", iq_p.data_ptr(), soft_p.data_ptr()))
            evs = {name: [] for name, _, _ in pairs}
            for i in range(2 + 6):                                # launches alternated: clocks and power drift hit both alike
                for name, a, b in pairs:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    ctx.ofdm_demod_frames_dd_dev(a + synth.NB_NULL * 8, synth.NB_FRAME_SAMPLES, n_frames, fo_cmp.data_ptr(), b,
                                                 cyc.data_ptr(), stream)
                    e1.record()
                    if i >= 2:
                        evs[name].append((e0, e1))
            torch.cuda.synchronize()
            res = {name: round(float(np.mean([x.elapsed_time(y) for x, y in v])), 3) for name, v in evs.items()}
            placement["front_end_ms_same_call_placed"] = res["placed"]
            placement["front_end_ms_same_call_plain_alloc"] = res["plain_alloc"]
            placement["plain_alloc_outputs_identical"] = bool(torch.equal(soft_p, soft))
            del iq_p, soft_p
        if not args.no_cp_leg:
            # The step as rounds 1-2 ran it: the caller asks for the cyclic-prefix correlations (the reference's estimator),
            # so the prefixes are read and the loop runs on them -- 21 % more bytes through the same kernel.
            torch.cuda.synchronize()
            soft_dd = soft.clone()                                # the timed run's last soft bits, for the comparison below
            evs = []
            t1 = time.perf_counter()
            for k in range(2 + args.steps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ctx.ofdm_demod_streams_dev(d_iq, synth.NB_FRAME_SAMPLES, E, F, BETA, soft.data_ptr(), cyc.data_ptr(), None, stream)
                e1.record()
                decode_into(soft, args.warmup + args.steps + k)
                if k == 1:
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                if k >= 2:
                    evs.append((e0, e1))
            torch.cuda.synchronize()
            cp_s = time.perf_counter() - t1
            cp_ms = float(np.mean([x.elapsed_time(y) for x, y in evs]))
            fib_c, crc_c, msc_c = fib.cpu().numpy(), crc.cpu().numpy(), msc.cpu().numpy()
            # the two loops sit a few 1e-5 carriers apart, so a few soft bits land on the other side of a truncation
            n_diff, max_diff = 0, 0
            for lo in range(0, n_frames, 1024):                   # in slices: the int16 difference of 3.8 GB at once is 7.5 GB
                dlt = (soft[lo:lo + 1024].to(torch.int16) - soft_dd[lo:lo + 1024].to(torch.int16)).abs()
                n_diff += int((dlt != 0).sum().item())
                max_diff = max(max_diff, int(dlt.max().item()))
            del soft_dd
            out["with_cyclic_prefix_correlations"] = {
                "value": n_frames * args.steps / cp_s, "unit": "frames/s", "ms_per_step": cp_s / args.steps * 1e3,
                "front_end_call_ms": cp_ms, "algorithmic_bytes_per_frame": A_OFDM_CP,
                "roofline_frac": A_OFDM_CP * n_frames / (cp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "outputs_identical_to_timed_run": bool((fib_c == fib_h).all() and (crc_c == crc_h).all() and (msc_c == msc_h).all()),
                "soft_bits_differing_from_timed_run_per_million": n_diff / (n_frames * dabgpu.NB_FRAME_BITS) * 1e6,
                "max_abs_soft_bit_difference": max_diff}
        if not args.no_sustained and args.sustained_seconds > 0:
            # The same step, repeated for >= --sustained-seconds: the 10-step timed region above lasts 0.1 s, shorter than
            # the package's power controller takes to settle (the front end runs at the 1400 W limit), so the
            # steady state gets a leg of its own.  Reported beside `value`, never as `value`.
            ms0 = elapsed / args.steps * 1e3
            n_sus = int(min(20000, max(args.steps, np.ceil(args.sustained_seconds * 1e3 / ms0))))
            ctx.set_timing(True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for k in range(n_sus):
                step(args.warmup + args.steps + k, False)
            torch.cuda.synchronize()
            sus_s = time.perf_counter() - t1
            sus_ofdm_ms, sus_launches = ctx.mean_kernel_ms(0)
            ctx.set_timing(False)
            fib_u, crc_u, msc_u = fib.cpu().numpy(), crc.cpu().numpy(), msc.cpu().numpy()
            sus_ach = A_OFDM * n_frames / (sus_ofdm_ms * 1e-3) / 1e9
            out["sustained"] = {"steps": n_sus, "seconds": sus_s, "ms_per_step": sus_s / n_sus * 1e3,
                                "value": n_frames * n_sus / sus_s, "unit": "frames/s",
                                "x_realtime": n_frames * n_sus / sus_s / REALTIME_FPS,
                                "front_end_kernel_ms": sus_ofdm_ms, "launches_timed": sus_launches,
                                "roofline_frac": sus_ach / HBM_PEAK_GBS,
                                "outputs_identical_to_timed_run": bool((fib_u == fib_h).all() and (crc_u == crc_h).all()
                                                                       and (msc_u == msc_h).all())}
        if not args.no_fft_stage:
            # the unfused FFT stage with the offsets the closed loop arrived at (per frame, from the stream states)
            fo = torch.from_numpy(np.repeat(net.astype(np.float32), F)).to(dev)
            # (45 % of this stage's traffic is the spectra it writes: the output buffer goes where it does not share
            # an HBM domain with the samples, dabgpu_device_alloc_apart)
            spec_bytes = n_frames * 76 * 2048 * 8
            d_spec, spec_probe = ctx.device_alloc_apart(spec_bytes, iq.data_ptr(), n_frames * synth.NB_FRAME_SAMPLES * 8)
            spectra = dabgpu.device_tensor(torch, d_spec, (n_frames, 76, 2048), torch.complex64, dev)
            evs = []
            for i in range(3 + 5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ctx.fft_symbols_dev(d_iq, synth.NB_FRAME_SAMPLES, n_frames, fo.data_ptr(), spectra.data_ptr(), stream)
                e1.record()
                if i >= 3:
                    evs.append((e0, e1))
            torch.cuda.synchronize()
            fft_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
            ach = A_FFT * n_frames / (fft_ms * 1e-3) / 1e9
            out["roofline_fft_stage"] = {"bound": "hbm", "kernel": "dabk::ofdm_wave_kernel<true,false> (FFT stage only)", "achieved": ach,
                                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                         "avg_launch_ms": fft_ms, "algorithmic_bytes_per_frame": A_FFT,
                                         # SURVEY 8(d)'s A_fft counts the cyclic prefixes; this kernel transforms the useful
                                         # 2048 samples of a symbol and never reads them: what it moves is 11 % less
                                         "bytes_moved_per_frame": A_FFT_MOVED,
                                         "frac_on_bytes_moved": A_FFT_MOVED * n_frames / (fft_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                         "output_placement": {"mover_ms_on_placed_buffer": round(spec_probe[0], 3),
                                                              "written_beside_same_domain_reads_per_mille": round(spec_probe[1], 1),
                                                              "classify_ms": round(spec_probe[2], 1)}}
            del spectra
            ctx.device_free(d_spec)
        if not args.no_selective:
            # The same step with the front end writing only what this workload decodes (FIC + the sub-channel,
            # dabgpu_ofdm_set_soft_selection): reported beside `value`, never as `value` -- the headline keeps the
            # reference's data flow (whole 230400-bit frames out of the demodulator).
            ctx.set_soft_selection(dabgpu.soft_selection([sc]))
            soft.zero_(); fib.zero_(); crc.zero_(); msc.zero_()
            n_before = len(ofdm_ev)
            for k in range(2):
                step(args.warmup + args.steps + k, False)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for k in range(args.steps):
                step(args.warmup + args.steps + 2 + k, True)
            torch.cuda.synchronize()
            sel_s = time.perf_counter() - t1
            ctx.set_soft_selection(None)
            fib_s, crc_s, msc_s = fib.cpu().numpy(), crc.cpu().numpy(), msc.cpu().numpy()
            sel_ok = bool(crc_s.all()) and bool((fib_s == fib_h).all()) and bool((msc_s == msc_h).all())
            sel_ofdm = float(np.mean([a.elapsed_time(b) for a, b in ofdm_ev[n_before:]]))
            sel = dabgpu.soft_selection([sc])
            kept = sum(c for _, c in sel)
            # symbols that are transformed: those carrying selected bits and their differential references; with the
            # decision-directed loop of the timed step the others are not read at all (a loop on the cyclic-prefix
            # correlations would read their prefix + last 512 samples)
            wanted = np.zeros(76, bool)
            for first, count in sel:
                wanted[1 + first // 3072: 1 + (first + count - 1) // 3072 + 1] = True
            need = wanted | np.append(wanted[1:], False)
            a_sel = int(need.sum()) * 2048 * 8 + kept
            out["selective_soft_output"] = {
                "value": n_frames * args.steps / sel_s, "unit": "frames/s", "ms_per_step": sel_s / args.steps * 1e3,
                "ofdm_avg_launch_ms": sel_ofdm, "soft_bits_written_per_frame": kept,
                "symbols_transformed_per_frame": int(need.sum()),
                "algorithmic_bytes_per_frame": a_sel,
                "ofdm_achieved_GBps": a_sel * n_frames / (sel_ofdm * 1e-3) / 1e9,
                "outputs_identical_to_whole_frame_run": sel_ok}
        if not args.no_closed_loop:
            out["closed_loop"] = closed_loop_leg(torch, dabgpu, synth, ctx, dev, stream, iq, ens, sc, soft, fib, crc, msc, hist,
                                                 E, F, args.steps)
        if world == 1 and args.cpu_seconds > 0:
            k = min(n_frames, 64)
            iq_h = iq.reshape(n_frames, -1)[:k, synth.NB_NULL:].contiguous().cpu().numpy()
            fo_h = np.repeat(-cfo_true, F)[:k].astype(np.float32)            # the oracle is handed the channel's offsets
            out["cpu_baseline"] = cpu_baseline(iq_h, fo_h, sc.length * 64, ens[0].mask, 64 * 24 + 6, args.cpu_seconds,
                                               len(os.sched_getaffinity(0)) or 1,
                                               truth_fibs=[ens[0].fibs[f % 4] for f in range(4)])
        print(json.dumps(out))
    d_bufs = (iq.data_ptr(), soft.data_ptr())
    del iq, soft
    ctx.free_frame_buffers(*d_bufs)
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def closed_loop_leg(torch, dabgpu, synth, ctx, dev, stream, iq, ens, sc, soft, fib, crc, msc, hist, E, F, steps):
    """The same samples as unaligned captures with nothing known about them: every ensemble's F frames form one
    capture that starts at an arbitrary sample.  Per step: dabgpu_acquire_dev (null-symbol search, per-frame fractional
    + whole-carrier frequency and timing from the PRS) -> dabgpu_ofdm_demod_acquired_dev (frames demodulated where
    they lie) -> dabgpu_decode_frames_dev.  Frames that are cut off at either end of a capture are not found, so a
    capture yields F-1 frames."""
    L = synth.NB_FRAME_SAMPLES
    rng = np.random.default_rng(0xACC)
    off = int(rng.integers(3000, L - 3000))                       # where the captures begin inside their first frame
    n_samples = F * L - off
    d_cap = iq.data_ptr() + off * 8
    acq = torch.zeros((E * F * 32,), dtype=torch.uint8, device=dev)
    counts = torch.zeros((E,), dtype=torch.int32, device=dev)
    for h in hist:
        h.zero_()
    soft.zero_(); fib.zero_(); crc.zero_(); msc.zero_()

    def step(k):
        ctx.acquire_dev(d_cap, F * L, E, n_samples, F, acq.data_ptr(), counts.data_ptr(), None, stream)
        ctx.ofdm_demod_acquired_dev(d_cap, F * L, E, F, acq.data_ptr(), soft.data_ptr(), None, None, stream)
        ctx.decode_frames_dev(soft.data_ptr(), dabgpu.NB_FRAME_BITS, E, F, fib.data_ptr(), crc.data_ptr(), [sc],
                              [None], [None], [msc.data_ptr()], stream)
    for k in range(2):
        step(k)
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    t0 = time.perf_counter()
    for k in range(steps):
        if k == steps - 1:
            evs[0].record()
            ctx.acquire_dev(d_cap, F * L, E, n_samples, F, acq.data_ptr(), counts.data_ptr(), None, stream)
            evs[1].record()
            ctx.ofdm_demod_acquired_dev(d_cap, F * L, E, F, acq.data_ptr(), soft.data_ptr(), None, None, stream)
            evs[2].record()
            ctx.decode_frames_dev(soft.data_ptr(), dabgpu.NB_FRAME_BITS, E, F, fib.data_ptr(), crc.data_ptr(), [sc],
                                  [None], [None], [msc.data_ptr()], stream)
            evs[3].record()
        else:
            step(k)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    def verify():
        """(frames found, frames locked, FIC bit-exact, MSC bit-exact) of what the last step left in the buffers"""
        cnt = counts.cpu().numpy()
        frames = acq.cpu().numpy().view(dabgpu.ACQUIRED_FRAME_DTYPE).reshape(E, F)
        fib_h, crc_h, msc_h = fib.cpu().numpy().reshape(E, F, 12, 32), crc.cpu().numpy().reshape(E, F, 12), msc.cpu().numpy()
        found = int(cnt.sum())
        locked = 0
        ok_fic, ok_msc = True, True
        for s in range(E):
            e = ens[s]
            for i in range(int(cnt[s])):
                fr = frames[s, i]
                if (fr["flags"] & 3) != 3:
                    continue
                locked += 1
                j = int(round((int(fr["start"]) + off - synth.NB_NULL) / L))         # which transmitted frame this is
                ok_fic &= bool(crc_h[s, i].all()) and bool((fib_h[s, i] == e.fibs[j % 4]).all())
                for c in range(4):
                    t = 4 * i + c                                                     # CIF index inside the capture
                    if t >= 15:                                                       # de-interleaver filled (no carried history)
                        ok_msc &= bool((msc_h[s, t] == e.msc_bytes[(4 * (j - i) + t - 15) % 16]).all())
        return found, locked, ok_fic and locked > 0, ok_msc and locked > 0

    found, locked, ok_fic, ok_msc = verify()
    out = {"value": locked * steps / el, "unit": "frames/s", "ms_per_step": el / steps * 1e3,
           "frames_found_per_step": found, "frames_locked_per_step": locked, "frames_in_the_captures": E * (F - 1),
           "acquire_ms": evs[0].elapsed_time(evs[1]), "ofdm_ms": evs[1].elapsed_time(evs[2]),
           "decode_ms": evs[2].elapsed_time(evs[3]),
           "fic_bit_exact": ok_fic, "msc_bit_exact": ok_msc,
           "what": "captures start %d samples into a frame; dabgpu_acquire_dev -> dabgpu_ofdm_demod_acquired_dev -> "
                   "dabgpu_decode_frames_dev; no offset, timing or alignment supplied" % off}

    # ---- tracking: the streams are acquired ONCE (step 0: the sequence above + dabgpu_track_start_dev); every later
    # capture goes through dabgpu_ofdm_demod_tracked_dev alone -- per frame a PRS synchronisation at the position the
    # stream's state predicts (no second pass over the capture for the null-symbol search), demodulation where the frame
    # lies, then the state update (fine-frequency loop, next frame start, drift).  The same buffer stands for the next
    # capture: it "begins" found-frames x 196608 samples later, which puts the next frame where this call's first one was.
    per_stream = found // E
    advance = per_stream * L
    soft.zero_(); fib.zero_(); crc.zero_(); msc.zero_()
    ctx.streams_reset(E)
    tcfg = dabgpu.track_cfg(auto_acquire=1)          # streams that are not tracking are acquired inside the call

    def tstep():
        ctx.ofdm_demod_tracked_dev(d_cap, F * L, E, n_samples, F, advance, soft.data_ptr(), acq.data_ptr(), counts.data_ptr(),
                                   tcfg, None, None, stream)
        ctx.decode_frames_dev(soft.data_ptr(), dabgpu.NB_FRAME_BITS, E, F, fib.data_ptr(), crc.data_ptr(), [sc],
                              [None], [None], [msc.data_ptr()], stream)
    tstep()                                          # the first call: every stream acquired (untimed, as in a receiver's life)
    torch.cuda.synchronize()
    assert all(ctx.get_stats(s).tracking == 1 for s in range(E)), "auto-acquisition did not lock every stream"
    for k in range(2):
        tstep()
    torch.cuda.synchronize()
    tev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t0 = time.perf_counter()
    for k in range(steps):
        if k == steps - 1:
            tev[0].record()
            ctx.ofdm_demod_tracked_dev(d_cap, F * L, E, n_samples, F, advance, soft.data_ptr(), acq.data_ptr(), counts.data_ptr(),
                                       tcfg, None, None, stream)
            tev[1].record()
            ctx.decode_frames_dev(soft.data_ptr(), dabgpu.NB_FRAME_BITS, E, F, fib.data_ptr(), crc.data_ptr(), [sc],
                                  [None], [None], [msc.data_ptr()], stream)
            tev[2].record()
        else:
            tstep()
    torch.cuda.synchronize()
    tel = time.perf_counter() - t0
    tfound, tlocked, tfic, tmsc = verify()
    stats = [ctx.get_stats(s) for s in range(E)]
    out["tracking"] = {"value": tlocked * steps / tel, "unit": "frames/s", "ms_per_step": tel / steps * 1e3,
                       "frames_found_per_step": tfound, "frames_locked_per_step": tlocked,
                       "track_sync_demod_update_ms": tev[0].elapsed_time(tev[1]), "decode_ms": tev[1].elapsed_time(tev[2]),
                       "fic_bit_exact": tfic, "msc_bit_exact": tmsc,
                       "streams_tracking": int(sum(st.tracking for st in stats)),
                       "frames_desync_total": int(sum(st.total_frames_desync for st in stats)),
                       "max_abs_drift_samples_per_frame": float(max(abs(st.drift) for st in stats)),
                       "auto_acquire": True,
                       "what": "ONE entry point from the first capture on: dabgpu_ofdm_demod_tracked_dev with cfg.auto_acquire -- the "
                               "first call (untimed) acquired every stream inside the call; every timed step: PRS synchronisation at "
                               "the predicted positions, demodulation in place, state update on the device (a stream that lost "
                               "lock would be re-acquired by the next call) -> dabgpu_decode_frames_dev"}
    return out


if __name__ == "__main__":
    main()
