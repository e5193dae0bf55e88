"""bench_legs.py -- everything bench.py measures AFTER its timed region, on rank 0, reported beside `value` and never as
`value`.  Each leg adds one key to the JSON line:

  with_decision_directed_loop      the timed step on the library's OWN fine-frequency estimator (opt-in: of a frame's 76
                                   cyclic prefixes only the PRS's is read, 17 % fewer bytes); also copied to the top level
                                   as value_own_estimator / roofline.frac_own_estimator
  sustained                        the timed step repeated for seconds (the package's power-limited steady state)
  roofline_fft_stage               the unfused FFT stage (north_star's 40 % bar, SURVEY 8d A_fft)
  selective_soft_output            the front end writing only what this workload decodes
  single_ensemble                  BASELINE configs 2 and 3: ONE ensemble, device-resident, and the plugin's real use -- one
                                   frame at a time from host memory through the two calls the host mirror makes
  host_fed                         BASELINE.md section 4 item 5: 64 frames per call from page-locked host memory, synchronous
                                   calls and the pipelined ring (dabgpu_pipe_*)
  host_mirror_end_to_end           the host mirror as the plugin runs it: dab_host_demo (OFDM_Demod::Process on one thread, the
                                   2-frame ring, BasicRadio::Process on the radio thread, FIG database, DAB+ channels) over a
                                   stream of three DAB+ services, samples in memory
  closed_loop                      the same samples as unaligned captures: acquisition every step, and tracking
  cpu_baseline                     the oracle and the SIMD port on the box's host cores (a bounded sample)
and `measure_traffic` replaces roofline.traffic (a figure from a tracked file) by one measured while the bench runs.
"""
import ctypes as C
import os
import time

import numpy as np

A_OFDM = 76 * 2552 * 8 + 230400            # 1 782 016 B: the timed step (the reference's data flow: every prefix is read)
A_OFDM_DD = (76 * 2048 + 504) * 8 + 230400 # 1 479 616 B: the decision-directed step (the PRS keeps its prefix: it picks the branch)
A_FFT = 76 * 2552 * 8 + 76 * 2048 * 8      # 2 796 800 B (unfused FFT stage, SURVEY 8(d): prefixes counted)
A_FFT_MOVED = 2 * 76 * 2048 * 8            # 2 490 368 B (what the FFT-stage kernel reads and writes)
HBM_PEAK_GBS = 8000.0
REALTIME_FPS = 1.0 / 0.096


def run(B, out):
    a = B.args
    if not a.no_traffic:
        t = measure_traffic(B.n_frames, os.path.dirname(os.path.abspath(__file__)), decoder=a.decoder)
        if t is not None and t.get("decoder") and out["decoder"].get("roofline"):
            decoder_traffic(out["decoder"]["roofline"], t["decoder"],
                            "measured in this run: the same two rocprofv3 --pmc child processes as roofline.traffic (the child decodes "
                            "the soft bits its front end left: noise in, every survivor path exercised)")
        if t is not None:
            out["roofline"]["traffic"] = t["hbm_bytes_per_launch"]
            out["roofline"]["traffic_over_algorithmic"] = t["ratio_to_algorithmic"]
            out["roofline"]["traffic_source"] = ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (one child process per "
                                                 "counter) around tools/pmc_traffic.py at this launch shape, counters calibrated on "
                                                 "that process's 1 GiB device copy (read scale %.4f, write scale %.4f)"
                                                 % (t["read_scale_from_1GiB_copy"], t["write_scale_from_1GiB_copy"]))
    if not a.no_dd_leg:
        leg = dd_leg(B)
        out["with_decision_directed_loop"] = leg
        out["value_own_estimator"] = leg["value"]
        out["roofline"]["frac_own_estimator"] = leg["roofline_frac"]
    if not a.no_sustained and a.sustained_seconds > 0:
        out["sustained"] = sustained_leg(B)
    if not a.no_fft_stage:
        out["roofline_fft_stage"] = fft_stage_leg(B)
    if not a.no_selective:
        out["selective_soft_output"] = selective_leg(B)
    if not a.no_single_ensemble:
        out["single_ensemble"] = single_ensemble_leg(B)
    if not a.no_host_fed:
        out["host_fed"] = host_fed_leg(B)
    if not a.no_host_mirror:
        out["host_mirror_end_to_end"] = host_mirror_leg(B)
    if not a.no_closed_loop:
        out["closed_loop"] = closed_loop_leg(B)
    if a.cpu_seconds > 0:
        k = min(B.n_frames, 64)
        iq_h = B.iq.reshape(B.n_frames, -1)[:k, B.synth.NB_NULL:].contiguous().cpu().numpy()
        fo_h = np.repeat(-B.cfo_true, B.F)[:k].astype(np.float32)            # the oracle is handed the channel's offsets
        out["cpu_baseline"] = cpu_baseline(iq_h, fo_h, B.sc.length * 64, B.ens[0].mask, 64 * 24 + 6, a.cpu_seconds,
                                           len(os.sched_getaffinity(0)) or 1,
                                           truth_fibs=[B.ens[0].fibs[f % 4] for f in range(4)])


def measure_traffic(n_frames, root, timeout_s=240, decoder="auto"):
    """HBM bytes per launch of the fused front end (the timed step's data flow: cyclic-prefix correlations out) from the TCC
    counters, measured NOW: two child
    processes `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 tools/pmc_traffic.py n cp` (counters cannot be
    read from inside a running process; one pass per counter, as MI355X_MICROARCH.md prescribes; the child does a 1 GiB
    device copy first, which calibrates the counters' unit).  Returns a dict, or None when the profiler is not there or a
    pass fails -- the caller then falls back to the tracked figure.  Never nested: when this process itself runs under a
    profiler (tools/prof.sh without --legs none), the child launcher would inherit the outer profiler's preload and the
    counters of a nested session mean nothing -- return None instead."""
    import csv
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof) or profiler_attached():
        return None
    out_dir = tempfile.mkdtemp(prefix="dabgpu_pmc_")
    env = {k: v for k, v in os.environ.items() if not is_profiler_variable(k, v)}
    env["TMPDIR"] = "/tmp"
    res = {}
    try:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(out_dir, c)
            r = subprocess.run([prof, "--pmc", c, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "t", "--",
                                "python3", os.path.join(root, "tools", "pmc_traffic.py"), str(n_frames), "cp", decoder],
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            path = os.path.join(d, "t_counter_collection.csv")
            if r.returncode != 0 or not os.path.exists(path):
                return None
            acc = {"ofdm": [], "copy": []}
            acc.update({k: [] for k in DECODER_KERNELS})
            for row in csv.DictReader(open(path)):
                if row["Counter_Name"] != c:
                    continue
                k = row["Kernel_Name"]
                v = float(row["Counter_Value"])
                if "ofdm_wave_kernel" in k:
                    acc["ofdm"].append(v)
                elif ("copy" in k.lower() or "clone" in k.lower()) and v > 1e5:
                    acc["copy"].append(v)
                else:
                    for dk in DECODER_KERNELS:
                        if dk in k:
                            acc[dk].append(v)
            if not acc["ofdm"] or not acc["copy"]:
                return None
            res[c] = {k: sum(v[-3:]) / len(v[-3:]) for k, v in acc.items() if v}
    except Exception:
        return None
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)
    gib = 1024 ** 3
    rd = res["FETCH_SIZE"]["ofdm"] * gib / res["FETCH_SIZE"]["copy"]          # the copy read 1 GiB and wrote 1 GiB
    wr = res["WRITE_SIZE"]["ofdm"] * gib / res["WRITE_SIZE"]["copy"]
    out = {"hbm_bytes_per_launch": rd + wr, "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
           "ratio_to_algorithmic": (rd + wr) / (A_OFDM * n_frames),
           "read_scale_from_1GiB_copy": gib / (res["FETCH_SIZE"]["copy"] * 1024), "write_scale_from_1GiB_copy": gib / (res["WRITE_SIZE"]["copy"] * 1024)}
    # the channel decoder's kernels, out of the same two passes (the child decodes the soft bits its front end left)
    if all(dk in res["FETCH_SIZE"] and dk in res["WRITE_SIZE"] for dk in DECODER_KERNELS):
        out["decoder"] = {dk: {"hbm_read_bytes_per_launch": res["FETCH_SIZE"][dk] * gib / res["FETCH_SIZE"]["copy"],
                               "hbm_write_bytes_per_launch": res["WRITE_SIZE"][dk] * gib / res["WRITE_SIZE"]["copy"]}
                          for dk in DECODER_KERNELS}
    return out


DECODER_KERNELS = ("lane_forward_grouped_kernel", "lane_traceback_grouped_kernel", "msc_history_kernel")
# static VALU count of the lane decoder's forward pass per trellis step of one wave (64 codewords), DESIGN.md 4.2b
LANE_VALU_PER_STEP, LANE_ACS_VALU_PER_STEP = 202, 96
# a packed-int16 VALU instruction of one wave occupies its SIMD for 4 cycles at saturation (profiles/r01_ubench_pk_rate.txt:
# v_pk_add_u16 / v_pk_max_i16 4.24, v_and / v_perm 3.94 with 4 waves per SIMD); 1024 SIMDs at 2.4 GHz
PK_I16_CYCLES, N_SIMDS, CLOCK_HZ = 4.0, 1024, 2.4e9
A_DECODER = 9216 + 396 + 37632             # A_fic + A_msc64 = 47 244 B per frame


def decoder_roofline(parts_ms, n_frames, codeword_steps):
    """The channel decoder's place against its two bounds, from the library's own HIP-event timers
    around the kernels of the timed steps (parts_ms = dabgpu_mean_kernel_ms 4 / 5 / 6: forward pass | traceback | history
    copy, + the launches they average over):
      forward pass -- VALU issue: wave-steps x instructions x 4 cycles / (1024 SIMDs x 2.4 GHz), priced on its 96 ACS
                      instructions alone (frac_of_acs_bound) and on all 202 (frac_of_valu_issue_bound);
      traceback    -- HBM: the survivor words it reads back (8 B per codeword-step), and the measured bytes once
                      decoder_traffic() has them.
    codeword_steps: [(codewords per launch, trellis steps)] of the launch's entries."""
    fwd_ms, tb_ms, hist_ms, n_fwd = parts_ms
    wave_steps = sum(((cw + 63) // 64) * steps for cw, steps in codeword_steps)
    acs_ms = wave_steps * LANE_ACS_VALU_PER_STEP * PK_I16_CYCLES / (N_SIMDS * CLOCK_HZ) * 1e3
    valu_ms = wave_steps * LANE_VALU_PER_STEP * PK_I16_CYCLES / (N_SIMDS * CLOCK_HZ) * 1e3
    survivor_bytes = sum(((cw + 63) // 64) * 64 * steps * 8 for cw, steps in codeword_steps)      # written once, read once
    return {"bound": "valu", "kernel": "dabk::lane_forward_grouped_kernel (forward pass; the traceback behind it is HBM-bound)",
            "valu_per_step": LANE_VALU_PER_STEP, "acs_valu_per_step": LANE_ACS_VALU_PER_STEP,
            "acs_share_of_valu": LANE_ACS_VALU_PER_STEP / LANE_VALU_PER_STEP,
            "wave_steps_per_launch": wave_steps, "cycles_per_packed_int16_instruction": PK_I16_CYCLES, "simds": N_SIMDS,
            "clock_GHz": CLOCK_HZ / 1e9,
            "acs_only_bound_ms": acs_ms, "valu_issue_bound_ms": valu_ms,
            "forward_ms": fwd_ms, "launches_timed": n_fwd,
            "frac_of_acs_bound": acs_ms / fwd_ms, "frac_of_valu_issue_bound": valu_ms / fwd_ms,
            "traceback_ms": tb_ms, "history_ms": hist_ms,
            "survivor_bytes_per_launch": survivor_bytes,
            "traceback_GBps_on_survivor_bytes": survivor_bytes / (tb_ms * 1e-3) / 1e9,
            "algorithmic_bytes_per_frame": A_DECODER, "frames_per_launch": n_frames,
            "traffic": None, "traffic_over_algorithmic": None, "traceback_GBps": None, "traffic_source": "not measured in this run"}


def decoder_traffic(r, traffic, source):
    """HBM bytes of the decoder's kernels (measure_traffic()["decoder"], or the tracked profiles/pmc_traffic_decoder.json)
    into the record decoder_roofline() made."""
    tot = sum(v["hbm_read_bytes_per_launch"] + v["hbm_write_bytes_per_launch"] for v in traffic.values())
    tb = traffic["lane_traceback_grouped_kernel"]
    tb_bytes = tb["hbm_read_bytes_per_launch"] + tb["hbm_write_bytes_per_launch"]
    r.update({"traffic": tot, "traffic_over_algorithmic": tot / (A_DECODER * r["frames_per_launch"]), "traffic_source": source,
              "traffic_by_kernel": {k: {"read": v["hbm_read_bytes_per_launch"], "write": v["hbm_write_bytes_per_launch"]} for k, v in traffic.items()},
              "traceback_GBps": tb_bytes / (r["traceback_ms"] * 1e-3) / 1e9,
              "traceback_frac_of_hbm_peak": tb_bytes / (r["traceback_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS})


def is_profiler_variable(k, v):
    """environment a rocprofv3 session plants in its target (and which a child of ours must not inherit)"""
    if k.startswith(("ROCP_", "ROCPROF", "ROCPROFILER_", "ROCTX_", "HSA_TOOLS_")):
        return True
    return k == "LD_PRELOAD" and ("rocprof" in v.lower() or "roctx" in v.lower())


def profiler_attached():
    return any(is_profiler_variable(k, v) for k, v in os.environ.items())


def _events(torch):
    return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def dd_leg(B):
    """The timed step on the library's OWN estimator (opt-in, dabgpu_set_stream_loop(decision_directed = 1)): the fine loop
    runs on the fourth powers of the differential symbols, its 0.2-carrier ambiguity resolved by the PRS's cyclic prefix --
    the only one of a frame's 76 prefixes that is still read: 17 % fewer bytes through the same kernel.  Reported beside
    `value`, never as `value`: the reference has no such loop.  The loop is switched back before the leg returns."""
    torch, ctx = B.torch, B.ctx
    torch.cuda.synchronize()
    soft_cp = B.soft.clone()                                # the timed run's last soft bits, for the comparison below
    ctx.set_stream_loop(decision_directed=True)
    steps = B.args.steps
    for k in range(2):                                      # the loop settles on the other estimator (untimed)
        B.step(B.args.warmup + steps + k, False)
    n_before = len(B.ofdm_ev)
    torch.cuda.synchronize()
    ctx.set_timing(True)
    t1 = time.perf_counter()
    for k in range(steps):
        B.step(B.args.warmup + steps + 2 + k, True)
    torch.cuda.synchronize()
    dd_s = time.perf_counter() - t1
    k_ms, k_launches = ctx.mean_kernel_ms(0)
    ctx.set_timing(False)
    call_ms = float(np.mean([x.elapsed_time(y) for x, y in B.ofdm_ev[n_before:]]))
    dec_ms = float(np.mean([x.elapsed_time(y) for x, y in B.dec_ev[n_before:]]))
    fib_c, crc_c, msc_c = B.fib.cpu().numpy(), B.crc.cpu().numpy(), B.msc.cpu().numpy()
    # the two loops sit a few 1e-5 carriers apart, so a few soft bits land on the other side of a truncation
    n_diff, max_diff = 0, 0
    for lo in range(0, B.n_frames, 1024):                   # in slices: the int16 difference of 3.8 GB at once is 7.5 GB
        dlt = (B.soft[lo:lo + 1024].to(torch.int16) - soft_cp[lo:lo + 1024].to(torch.int16)).abs()
        n_diff += int((dlt != 0).sum().item())
        max_diff = max(max_diff, int(dlt.max().item()))
    del soft_cp
    # its own ceiling: the mover of the same geometry without the 75 prefixes (leaves meaningless bytes in `soft`)
    mev = []
    for i in range(2 + 5):
        e0, e1 = _events(torch)
        e0.record()
        ctx.mover_frames_dev(B.d_iq, B.L, B.n_frames, B.soft.data_ptr(), False, B.stream)
        e1.record()
        if i >= 2:
            mev.append((e0, e1))
    torch.cuda.synchronize()
    mover_ms = float(np.mean([x.elapsed_time(y) for x, y in mev]))
    ctx.set_stream_loop(decision_directed=False)            # back to the library's default for every later leg
    for k in range(2):
        B.step(B.args.warmup + 2 * steps + 2 + k, False)
    torch.cuda.synchronize()
    return {"value": B.n_frames * steps / dd_s, "unit": "frames/s", "ms_per_step": dd_s / steps * 1e3,
            "x_realtime": B.n_frames * steps / dd_s / REALTIME_FPS,
            "front_end_call_ms": call_ms, "front_end_kernel_ms": k_ms, "launches_timed": k_launches, "decoder_ms": dec_ms,
            "algorithmic_bytes_per_frame": A_OFDM_DD,
            "roofline_frac": A_OFDM_DD * B.n_frames / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "mover_same_geometry_ms": mover_ms, "box_mover_frac": A_OFDM_DD * B.n_frames / (mover_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "kernel_over_mover": k_ms / mover_ms,
            "outputs_identical_to_timed_run": bool((fib_c == B.fib_h).all() and (crc_c == B.crc_h).all() and (msc_c == B.msc_h).all()),
            "soft_bits_differing_from_timed_run_per_million": n_diff / (B.n_frames * B.dabgpu.NB_FRAME_BITS) * 1e6,
            "max_abs_soft_bit_difference": max_diff,
            "what": "dabgpu_set_stream_loop(decision_directed = 1): this library's own estimator, not the reference's"}


def sustained_leg(B):
    """The same step, repeated for >= --sustained-seconds: the 10-step timed region lasts 0.1 s, shorter than the package's
    power controller takes to settle (the front end runs at the 1400 W limit), so the steady state gets a leg
    of its own."""
    torch, ctx, a = B.torch, B.ctx, B.args
    n_sus = int(min(20000, max(a.steps, np.ceil(a.sustained_seconds * 1e3 / B.ms_per_step))))
    ctx.set_timing(True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for k in range(n_sus):
        B.step(a.warmup + a.steps + k, False)
    torch.cuda.synchronize()
    sus_s = time.perf_counter() - t1
    sus_ofdm_ms, sus_launches = ctx.mean_kernel_ms(0)
    ctx.set_timing(False)
    fib_u, crc_u, msc_u = B.fib.cpu().numpy(), B.crc.cpu().numpy(), B.msc.cpu().numpy()
    return {"steps": n_sus, "seconds": sus_s, "ms_per_step": sus_s / n_sus * 1e3,
            "value": B.n_frames * n_sus / sus_s, "unit": "frames/s", "x_realtime": B.n_frames * n_sus / sus_s / REALTIME_FPS,
            "front_end_kernel_ms": sus_ofdm_ms, "launches_timed": sus_launches,
            "roofline_frac": A_OFDM * B.n_frames / (sus_ofdm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "outputs_identical_to_timed_run": bool((fib_u == B.fib_h).all() and (crc_u == B.crc_h).all() and (msc_u == B.msc_h).all())}


def fft_stage_leg(B):
    """the unfused FFT stage with the offsets the closed loop arrived at (per frame, from the stream states)"""
    torch, ctx = B.torch, B.ctx
    fo = torch.from_numpy(np.repeat(B.net.astype(np.float32), B.F)).to(B.dev)
    spectra = torch.empty((B.n_frames, 76, 2048), dtype=torch.complex64, device=B.dev)
    evs = []
    for i in range(3 + 5):
        e0, e1 = _events(torch)
        e0.record()
        ctx.fft_symbols_dev(B.d_iq, B.L, B.n_frames, fo.data_ptr(), spectra.data_ptr(), B.stream)
        e1.record()
        if i >= 3:
            evs.append((e0, e1))
    torch.cuda.synchronize()
    fft_ms = float(np.mean([x.elapsed_time(y) for x, y in evs]))
    del spectra
    ach = A_FFT * B.n_frames / (fft_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "dabk::ofdm_wave_kernel<true,false> (FFT stage only)", "achieved": ach,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "avg_launch_ms": fft_ms,
            "algorithmic_bytes_per_frame": A_FFT,
            # SURVEY 8(d)'s A_fft counts the cyclic prefixes; this kernel transforms the useful 2048 samples of a symbol
            # and never reads them: what it moves is 11 % less
            "bytes_moved_per_frame": A_FFT_MOVED,
            "frac_on_bytes_moved": A_FFT_MOVED * B.n_frames / (fft_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "output_buffer": "plain allocation"}


def selective_leg(B):
    """The same step with the front end writing only what this workload decodes (FIC + the sub-channel,
    dabgpu_ofdm_set_soft_selection): the headline keeps the reference's data flow (whole 230400-bit frames out of the
    demodulator)."""
    torch, ctx, a, dabgpu = B.torch, B.ctx, B.args, B.dabgpu
    sel = dabgpu.soft_selection([B.sc])
    ctx.set_soft_selection(sel)
    # (both are this library's own options, so they go together here: on the decision-directed loop a symbol nobody wants is
    # not read at all; on the reference's loop its cyclic prefix would still be correlated)
    ctx.set_stream_loop(decision_directed=True)
    B.soft.zero_(); B.fib.zero_(); B.crc.zero_(); B.msc.zero_()
    n_before = len(B.ofdm_ev)
    for k in range(2):
        B.step(a.warmup + a.steps + k, False)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for k in range(a.steps):
        B.step(a.warmup + a.steps + 2 + k, True)
    torch.cuda.synchronize()
    sel_s = time.perf_counter() - t1
    ctx.set_soft_selection(None)
    ctx.set_stream_loop(decision_directed=False)
    fib_s, crc_s, msc_s = B.fib.cpu().numpy(), B.crc.cpu().numpy(), B.msc.cpu().numpy()
    sel_ok = bool(crc_s.all()) and bool((fib_s == B.fib_h).all()) and bool((msc_s == B.msc_h).all())
    sel_ofdm = float(np.mean([x.elapsed_time(y) for x, y in B.ofdm_ev[n_before:]]))
    sel_dec = [x.elapsed_time(y) for x, y in B.dec_ev[n_before:]]
    kept = sum(c for _, c in sel)
    # symbols that are transformed: those carrying selected bits and their differential references; with the
    # decision-directed loop of the timed step the others are not read at all
    wanted = np.zeros(76, bool)
    for first, count in sel:
        wanted[1 + first // 3072: 1 + (first + count - 1) // 3072 + 1] = True
    need = wanted | np.append(wanted[1:], False)
    a_sel = int(need.sum()) * 2048 * 8 + kept
    return {"value": B.n_frames * a.steps / sel_s, "unit": "frames/s", "ms_per_step": sel_s / a.steps * 1e3,
            "ofdm_avg_launch_ms": sel_ofdm, "decoder_ms": float(np.mean(sel_dec)), "decoder_ms_max": float(np.max(sel_dec)),
            "soft_bits_written_per_frame": kept,
            "symbols_transformed_per_frame": int(need.sum()), "algorithmic_bytes_per_frame": a_sel,
            "ofdm_achieved_GBps": a_sel * B.n_frames / (sel_ofdm * 1e-3) / 1e9,
            "outputs_identical_to_whole_frame_run": sel_ok,
            "estimator": "decision-directed (opt-in, as the selection itself)"}


def single_ensemble_leg(B):
    """BASELINE configs 2 and 3 -- ONE ensemble on the GPU -- two ways.
    device_resident: the ensemble's F frames in HBM, one stream call + one decode call per step, OFDM + FIC (config 2) and
      OFDM + FIC + the 64 kbps sub-channel (config 3).  One stream cannot fill 256 CUs for long: this is a latency-bound
      shape, reported as it is.
    host_fed_per_frame: what the plugin does (/root/reference/src/dab_module.cpp:20-28 -> OFDM_Demod::Process;
      src/radio_block.cpp:33-44 -> BasicRadio::Process): one frame of IQ from (page-locked) host memory through
      dabgpu_ofdm_demod_stream_frame (synchronisation on the PRS + demodulation + loops on the device, one upload, one
      download), its soft bits through dabgpu_decode_stream_frames (FIC + the sub-channel, de-interleaver on the device);
      two contexts, as the host mirror holds them."""
    torch, dabgpu, synth = B.torch, B.dabgpu, B.synth
    F, L, dev, stream = B.F, B.L, B.dev, B.stream
    steps = max(B.args.steps, 5)
    st0 = B.ctx.get_stats(0)
    c1 = dabgpu.Context(device=B.dev_index, max_frames=F)
    c1.streams_reset(1)
    c1.set_stream_offsets(0, fine=float(st0.fine_freq_offset), coarse=float(st0.coarse_freq_offset))   # acquisition is untimed, as above
    soft, fib, crc, msc = B.soft[:F], B.fib[:F], B.crc[:F], B.msc[:1]
    hist = [h[:1] for h in B.hist]
    for h in hist:
        h.zero_()
    e = B.ens[0]
    res = {}
    # the rows BASELINE names run the library's defaults (the reference's estimator); `_own_estimator`: the same on the
    # opt-in decision-directed loop
    for name, scs, dd in (("ofdm_fic", [], False), ("ofdm_fic_msc64", [B.sc], False),
                          ("ofdm_fic_own_estimator", [], True), ("ofdm_fic_msc64_own_estimator", [B.sc], True)):
        fib.zero_(); crc.zero_(); msc.zero_()
        c1.set_stream_loop(decision_directed=dd)

        def one(k):
            c1.ofdm_demod_streams_dev(B.d_iq, L, 1, F, B.BETA, soft.data_ptr(), None, None, stream)
            c1.decode_frames_dev(soft.data_ptr(), dabgpu.NB_FRAME_BITS, 1, F, fib.data_ptr(), crc.data_ptr(), scs,
                                 [hist[k & 1].data_ptr()] if scs else [], [hist[(k & 1) ^ 1].data_ptr()] if scs else [],
                                 [msc.data_ptr()] if scs else [], stream)
        for k in range(2):
            one(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            one(2 + k)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        fib_h, crc_h = fib.cpu().numpy(), crc.cpu().numpy()
        ok = bool(crc_h.all()) and all(bool((fib_h[f] == e.fibs[f % 4]).all()) for f in range(F))
        row = {"value": F * steps / dt, "unit": "frames/s", "x_realtime": F * steps / dt / REALTIME_FPS,
               "ms_per_step": dt / steps * 1e3, "frames_per_step": F, "fic_bit_exact": ok}
        if scs:
            msc_h = msc.cpu().numpy()
            row["msc_bit_exact"] = all(bool((msc_h[0, t] == e.msc_bytes[(t - 15) % 16]).all()) for t in range(F * 4))
        # device time of the two calls of a step (events on the stream; after the checks: these extra calls move the loop
        # and the de-interleaver on)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        t_fe = t_dec = 0.0
        for k in range(steps):
            ev[0].record()
            c1.ofdm_demod_streams_dev(B.d_iq, L, 1, F, B.BETA, soft.data_ptr(), None, None, stream)
            ev[1].record()
            c1.decode_frames_dev(soft.data_ptr(), dabgpu.NB_FRAME_BITS, 1, F, fib.data_ptr(), crc.data_ptr(), scs,
                                 [hist[k & 1].data_ptr()] if scs else [], [hist[(k & 1) ^ 1].data_ptr()] if scs else [],
                                 [msc.data_ptr()] if scs else [], stream)
            ev[2].record()
            torch.cuda.synchronize()
            t_fe += ev[0].elapsed_time(ev[1]); t_dec += ev[1].elapsed_time(ev[2])
        row["front_end_call_ms"], row["decode_call_ms"] = t_fe / steps, t_dec / steps
        res[name] = row
    c1.close()

    # ---- one frame at a time from host memory ----
    M = 64                                                   # samples the host assembles its frames early (timing_margin)
    n_host = min(F - 1, 32)
    flat = B.iq[0].reshape(-1)
    used = 76 * 2552
    p_iq = dabgpu.PinnedArray((n_host, used), np.complex64)
    for f in range(n_host):
        lo = f * L + synth.NB_NULL - M
        p_iq.array[f] = flat[lo:lo + used].cpu().numpy()
    p_soft = dabgpu.PinnedArray((dabgpu.NB_FRAME_BITS,), np.int8)
    p_fib = dabgpu.PinnedArray((1, 12, 32), np.uint8)
    p_ok = dabgpu.PinnedArray((1, 12), np.uint8)
    p_out = dabgpu.PinnedArray((4, 192), np.uint8)
    lib = dabgpu.lib()
    co, cd = dabgpu.Context(device=B.dev_index, max_frames=1), dabgpu.Context(device=B.dev_index, max_frames=1)
    co.streams_reset(1)
    cfg = dabgpu.track_cfg(timing_margin=M)
    fres = dabgpu.FrameResult()
    arr = (dabgpu.Subchannel * 1)(B.sc)
    outs = (C.c_void_p * 1)(p_out.array.ctypes.data)
    t_ofdm, t_dec, ok_fic, ok_msc, ok_lock = [], [], True, True, True
    for rep in range(3):                                     # pass 0 warms up (and acquires); passes 1-2 are timed
        for f in range(n_host):
            t0 = time.perf_counter()
            rc = lib.dabgpu_ofdm_demod_stream_frame(co._h, 0, p_iq.array[f].ctypes.data, 1 if (rep == 0 and f == 0) else 0,
                                                    C.byref(cfg), p_soft.array.ctypes.data, None, C.byref(fres))
            t1 = time.perf_counter()
            rc2 = lib.dabgpu_decode_stream_frames(cd._h, p_soft.array.ctypes.data, dabgpu.NB_FRAME_BITS, 1, p_fib.array.ctypes.data,
                                                  p_ok.array.ctypes.data, arr, 1, outs)
            t2 = time.perf_counter()
            assert rc == 0 and rc2 == 0, (rc, rc2)
            if rep > 0:
                t_ofdm.append(t1 - t0); t_dec.append(t2 - t1)
            ok_lock &= fres.flags == 3
            ok_fic &= bool(p_ok.array.all()) and bool((p_fib.array[0] == e.fibs[f % 4]).all())
            g = rep * n_host + f                             # frames fed to the decoder so far (its de-interleaver continues)
            if rep == 0 and f >= 4:                          # (the passes after the first restart the multiplex mid-ring)
                for cif in range(4):
                    ok_msc &= bool((p_out.array[cif] == e.msc_bytes[(4 * g + cif - 15) % 16]).all())
        if rep == 0:
            cd.decode_stream_reset()
    co.close(); cd.close()
    for p in (p_iq, p_soft, p_fib, p_ok, p_out):
        p.close()
    o_ms, d_ms = float(np.mean(t_ofdm)) * 1e3, float(np.mean(t_dec)) * 1e3
    res["host_fed_per_frame"] = {
        "ofdm_demod_stream_frame_ms": o_ms, "decode_stream_frames_ms": d_ms, "frame_ms": o_ms + d_ms,
        "x_realtime_one_thread": 96.0 / (o_ms + d_ms), "x_realtime_two_threads_as_the_plugin": 96.0 / max(o_ms, d_ms),
        "frames_timed": len(t_ofdm), "every_frame_locked": bool(ok_lock), "fic_bit_exact": bool(ok_fic), "msc_bit_exact": bool(ok_msc),
        "what": "one frame (1.55 MB cf32) per call from page-locked host memory: dabgpu_ofdm_demod_stream_frame -> "
                "dabgpu_decode_stream_frames (FIC + one 64 kbps sub-channel), wall clock of each call on the host"}
    return res


def host_fed_leg(B):
    """IQ that starts in host memory.  64 consecutive frames of one stream per call from
    page-locked buffers, closed loop on the device, FIC + the sub-channel decoded, every result (soft bits included) back
    in host memory: the two synchronous calls one after the other, then the ring (dabgpu_pipe_*: upload of batch k+1,
    kernels of batch k and download of batch k-1 at the same time)."""
    torch, dabgpu, synth = B.torch, B.dabgpu, B.synth
    n = min(64, B.F)
    used = 76 * 2552
    e = B.ens[0]
    st0 = B.ctx.get_stats(0)
    p_iq = dabgpu.PinnedArray((n, used), np.complex64)
    p_iq.array[:] = B.iq[0, :n, synth.NB_NULL:synth.NB_NULL + used].cpu().numpy()
    S = 3
    sets = [dict(soft=dabgpu.PinnedArray((n, dabgpu.NB_FRAME_BITS), np.int8), fib=dabgpu.PinnedArray((n, 12, 32), np.uint8),
                 ok=dabgpu.PinnedArray((n, 12), np.uint8), out=dabgpu.PinnedArray((1, n * 4, 192), np.uint8)) for _ in range(S + 1)]
    c = dabgpu.Context(device=B.dev_index, max_frames=n)
    c.streams_reset(1)
    c.set_stream_offsets(0, fine=float(st0.fine_freq_offset), coarse=float(st0.coarse_freq_offset))
    lib = dabgpu.lib()
    arr = (dabgpu.Subchannel * 1)(B.sc)
    iq_bytes = n * used * 8

    # ---- synchronous: dabgpu_ofdm_demod_streams + dabgpu_decode_frames (the soft bits cross the link twice more) ----
    s0 = sets[0]
    h_in = dabgpu.PinnedArray((1, 15, B.sc.length * 64), np.int8)
    h_out = dabgpu.PinnedArray((1, 15, B.sc.length * 64), np.int8)
    h_in.array[:] = 0

    def sync_call():
        assert lib.dabgpu_ofdm_demod_streams(c._h, p_iq.array.ctypes.data, used, 1, n, C.c_float(B.BETA), s0["soft"].array.ctypes.data,
                                             None, None) == 0
        hi = (C.c_void_p * 1)(h_in.array.ctypes.data); ho = (C.c_void_p * 1)(h_out.array.ctypes.data)
        outs = (C.c_void_p * 1)(s0["out"].array.ctypes.data)
        assert lib.dabgpu_decode_frames(c._h, s0["soft"].array.ctypes.data, dabgpu.NB_FRAME_BITS, 1, n, s0["fib"].array.ctypes.data,
                                        s0["ok"].array.ctypes.data, arr, 1, hi, ho, outs) == 0
    for _ in range(3):
        sync_call()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        sync_call()
    sync_s = (time.perf_counter() - t0) / reps
    fib_ref = s0["fib"].array.copy()
    sync_ok = bool(s0["ok"].array.all()) and all(bool((fib_ref[f] == e.fibs[f % 4]).all()) for f in range(n))

    # ---- the ring ----
    c.pipe_open(S, n, used)
    calls = 120
    tickets = [None] * (S + 1)

    def ring(n_calls, want_soft):
        for k in range(n_calls):
            st = sets[k % (S + 1)]
            if tickets[k % (S + 1)] is not None:
                c.pipe_wait(tickets[k % (S + 1)])            # this set's previous batch must be home before it is reused
            tickets[k % (S + 1)] = c.pipe_submit(p_iq.array, 1, n, None, [B.sc], st["soft"].array if want_soft else None,
                                                 st["fib"].array, st["ok"].array, [st["out"].array], beta=B.BETA)
        for i in range(S + 1):
            if tickets[i] is not None:
                c.pipe_wait(tickets[i])
                tickets[i] = None
    ring(8, True)
    t0 = time.perf_counter()
    ring(calls, True)
    ring_s = (time.perf_counter() - t0) / calls
    ring_ok = all(bool(st["ok"].array.all()) and bool((st["fib"].array == fib_ref).all()) for st in sets)
    # the same 64 frames again and again: the multiplex is 16 CIFs long and a batch 256, so every batch continues the stream
    msc_ok = all(bool((sets[0]["out"].array[0, t] == e.msc_bytes[(t - 15) % 16]).all()) for t in range(n * 4))
    t0 = time.perf_counter()
    ring(calls, False)
    ring_nosoft_s = (time.perf_counter() - t0) / calls
    c.pipe_close()
    c.close()
    for st in sets:
        for p in st.values():
            p.close()
    for p in (p_iq, h_in, h_out):
        p.close()
    # the link itself: the same 99 MB upload, back to back, nothing else
    d = torch.empty(iq_bytes, dtype=torch.uint8, device=B.dev)
    h = torch.empty(iq_bytes, dtype=torch.uint8).pin_memory()
    for _ in range(2):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    h2d_gbps = iq_bytes * 10 / (time.perf_counter() - t0) / 1e9
    del d, h

    def row(sec):
        return {"value": n / sec, "unit": "frames/s", "ms_per_call": sec * 1e3, "iq_GBps_over_the_link": iq_bytes / sec / 1e9,
                "x_realtime": n / sec / REALTIME_FPS, "fraction_of_h2d_rate": iq_bytes / sec / 1e9 / h2d_gbps}
    out = {"frames_per_call": n, "h2d_copy_GBps": h2d_gbps,
           "synchronous_calls": dict(row(sync_s), fic_bit_exact=sync_ok,
                                     what="dabgpu_ofdm_demod_streams + dabgpu_decode_frames, page-locked buffers"),
           "ring": dict(row(ring_s), slots=S, fic_bit_exact=ring_ok, msc_bit_exact=msc_ok, outputs_identical_to_synchronous_calls=ring_ok,
                        what="dabgpu_pipe_submit / dabgpu_pipe_wait, soft bits + FIBs + sub-channel bytes back in host memory"),
           "ring_without_soft_bit_download": row(ring_nosoft_s)}
    return out


def host_mirror_leg(B, n_frames=400):
    """The C++ host mirror end to end, in a process of its own: `dab_host_demo` wires OFDM_Demod -> ThreadedRingBuffer ->
    radio thread -> BasicRadio exactly as /root/reference/src/radio_block.cpp:11-49 does and is fed in chunks of 65 536
    samples as /root/reference/src/dab_module.cpp:20-28 feeds it; the multiplex (three DAB+ services, 64 / 48 / 32 kbit/s)
    is unknown to it: the FIC is parsed, the services opened, their sub-channels decoded, super-frames checked (Fire code,
    RS(120,110), AU CRCs) and access units handed out.  The samples are read into memory first (DAB_DEMO_PRELOAD: SDR++ hands
    over blocks in memory); timed: the two threads from the first chunk to the last access unit."""
    import subprocess
    import tempfile
    synth = B.synth
    root = os.path.dirname(os.path.abspath(__file__))
    exe = os.path.join(root, "sdrplusplus-dab-radio-plugin_amd", "host", "dab_host_demo")
    if not os.path.exists(exe):
        return "not available: %s is not built" % exe
    services = [("Radio One", 0xC221, 3, 0, 3, 64, 0), ("Jazz 24", 0xC222, 7, 0, 2, 48, 48), ("News", 0xC223, 9, 1, 2, 32, 200)]
    ens = synth.ServiceEnsemble(1, services, n_frames=5)
    base = synth.channel(np.tile(ens.iq().ravel(), 4), snr_db=20.0, cfo=1.2 / 2048, rng=np.random.default_rng(8)).astype(np.complex64)
    reps = (n_frames + 19) // 20
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "iq.cf32")
        with open(path, "wb") as f:
            f.write(base[-30000:].tobytes())
            for _ in range(reps):
                f.write(base.tobytes())                          # 20 frames per repeat: the multiplex continues seamlessly
            f.write(base[:synth.NB_NULL + 5000].tobytes())
        r = None
        for _ in range(2):                                       # (the first run warms the file cache and the device)
            r = subprocess.run([exe, path, os.path.join(d, "out"), "65536"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                               timeout=300, env=dict(os.environ, DAB_DEMO_PRELOAD="1"))
            if r.returncode != 0:
                return "not available: dab_host_demo returned %d" % r.returncode
        db = open(os.path.join(d, "out.db")).read()
    kv = dict(t.split("=") for t in r.stdout.split() if "=" in t)
    frames, dt = int(kv["frames_read"]), float(kv["processing_s"])
    chans = [l for l in db.splitlines() if l.startswith("channel")]
    return {"value": frames / dt, "unit": "frames/s", "x_realtime": frames / dt / REALTIME_FPS, "us_per_frame": dt / frames * 1e6,
            "frames": frames, "frames_desync": int(kv["frames_desync"]), "fibs": int(kv["fibs"]), "fib_errors": int(kv["fib_errors"]),
            "seconds_in_ofdm_process": float(kv["ofdm_process_s"]), "seconds_in_radio_process": float(kv["radio_process_s"]),
            "dab_plus_services_decoded": len(chans),
            "all_superframes_clean": bool(chans) and all("firecode_error=0" in l and "rs_error=0" in l for l in chans),
            "what": "dab_host_demo, samples in memory, chunks of 65536: OFDM_Demod::Process on one thread, BasicRadio::Process on "
                    "the radio thread (FIC -> database -> three DAB+ services -> access units)"}


def closed_loop_leg(B):
    """The same samples as unaligned captures with nothing known about them: every ensemble's F frames form one
    capture that starts at an arbitrary sample.  Per step: dabgpu_acquire_dev (null-symbol search, per-frame fractional
    + whole-carrier frequency and timing from the PRS) -> dabgpu_ofdm_demod_acquired_dev (frames demodulated where
    they lie) -> dabgpu_decode_frames_dev.  Frames that are cut off at either end of a capture are not found, so a
    capture yields F-1 frames.  Then `tracking`: the streams are acquired ONCE and followed on the device."""
    torch, dabgpu, synth, ctx, dev, stream = B.torch, B.dabgpu, B.synth, B.ctx, B.dev, B.stream
    iq, ens, sc, soft, fib, crc, msc, hist, E, F, steps = B.iq, B.ens, B.sc, B.soft, B.fib, B.crc, B.msc, B.hist, B.E, B.F, B.args.steps
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

    def decode():
        ctx.decode_frames_dev(soft.data_ptr(), dabgpu.NB_FRAME_BITS, E, F, fib.data_ptr(), crc.data_ptr(), [sc],
                              [None], [None], [msc.data_ptr()], stream)

    def step(k):
        ctx.acquire_dev(d_cap, F * L, E, n_samples, F, acq.data_ptr(), counts.data_ptr(), None, stream)
        ctx.ofdm_demod_acquired_dev(d_cap, F * L, E, F, acq.data_ptr(), soft.data_ptr(), None, None, stream)
        decode()
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
            decode()
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
                   "dabgpu_decode_frames_dev; no offset, timing or alignment supplied.  Re-acquiring every step re-reads the "
                   "whole capture: what a receiver does ONCE; `tracking` is the steady state" % off}

    # ---- tracking: one entry point from the first capture on (auto_acquire); every later capture: per frame a PRS
    # synchronisation at the position the stream's state predicts, demodulation where the frame lies, then the state
    # update.  The same buffer stands for the next capture: it "begins" found-frames x 196608 samples later.
    per_stream = found // E
    advance = per_stream * L
    soft.zero_(); fib.zero_(); crc.zero_(); msc.zero_()
    ctx.streams_reset(E)
    tcfg = dabgpu.track_cfg(auto_acquire=1)          # streams that are not tracking are acquired inside the call

    def tracked():
        ctx.ofdm_demod_tracked_dev(d_cap, F * L, E, n_samples, F, advance, soft.data_ptr(), acq.data_ptr(), counts.data_ptr(),
                                   tcfg, None, None, stream)
    tracked(); decode()                              # the first call: every stream acquired (untimed, as in a receiver's life)
    torch.cuda.synchronize()
    assert all(ctx.get_stats(s).tracking == 1 for s in range(E)), "auto-acquisition did not lock every stream"
    for k in range(2):
        tracked(); decode()
    torch.cuda.synchronize()
    tev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t0 = time.perf_counter()
    for k in range(steps):
        if k == steps - 1:
            tev[0].record()
        tracked()
        if k == steps - 1:
            tev[1].record()
        decode()
        if k == steps - 1:
            tev[2].record()
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
                       "what": "dabgpu_ofdm_demod_tracked_dev with cfg.auto_acquire -> dabgpu_decode_frames_dev"}
    return out


# ------------------------------------------------------------------------------------------------ CPU baseline
def fftw_fft_stage(iq_host, seconds=3.0):
    """if the box happens to have FFTW3f (the reference's FFT library,
    /root/reference/CMakeLists.txt:55-64), time its 2048-point c2c transform on the FFT stage's work (76 per frame,
    one thread, FFTW_MEASURE).  Returns frames/s or None when the library is not installed."""
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
      value                    `threads` threads, one frame stream each; `effective_cores` = value / single_core_value says
                               how many cores' worth of time the box actually granted"""
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
    `threads` = the logical CPUs the scheduler lists, capped at what the cgroup quota grants (round 3 started 256 pthreads
    under a 16-core quota).  Stated baselines, never the target."""
    from oracle import oracle as O
    quota = cpu_quota_cores()
    if quota is not None:
        threads = max(1, min(threads, int(np.ceil(quota))))
    fftw = fftw_fft_stage(iq_host)
    out = cpu_rows(O, iq_host, fo_host, sc_len_bits, mask, nsteps, budget_s * 0.5, threads, False)
    out.update({"kind": "port", "cgroup_cpu_quota_cores": quota, "logical_cpus": len(os.sched_getaffinity(0)) or 1,
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
