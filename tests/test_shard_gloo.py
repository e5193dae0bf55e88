"""N>1 path on CPU: two gloo ranks shard the ensembles and run the report reductions
bench.py uses (max elapsed, sum frames, min flags).  No data-path collective exists."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dabgpu.shard import ensembles_of_rank, reduce_report, gather_per_rank


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = ensembles_of_rank(128, world, rank)
    elapsed, frames, flags = reduce_report(dist, torch.device("cpu"), 1.0 + rank, len(mine) * 16,
                                           [True, rank == 0])
    rows = gather_per_rank(dist, torch.device("cpu"), [10.0 * rank, rank + 0.5])
    dist.barrier()
    q.put((rank, mine, elapsed, frames, flags, rows))
    dist.destroy_process_group()


def test_two_rank_sharding_and_report():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    owned = res[0][1] + res[1][1]
    assert sorted(owned) == list(range(128)) and not set(res[0][1]) & set(res[1][1])
    for _, _, elapsed, frames, flags, rows in res:
        assert elapsed == 2.0 and frames == 128 * 16 and flags == [True, False]
        assert rows == [[0.0, 0.5], [10.0, 1.5]]                      # every rank sees every rank's row, in rank order


def test_single_process_report_passthrough():
    assert reduce_report(None, None, 0.5, 10, [True]) == (0.5, 10, [True])
    assert ensembles_of_rank(10, 4, 3) == [3, 7]
    assert gather_per_rank(None, None, [1, 2.5]) == [[1.0, 2.5]]


import json
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(nproc, extra_env=None, ensembles=4, frames=16, more=()):
    env = dict(os.environ)
    env.update(extra_env or {})
    args = ["--ensembles", str(ensembles), "--frames", str(frames), "--steps", "2", "--warmup", "1", "--legs", "none"] + list(more)
    if nproc == 1 and not (extra_env or {}).get("DABGPU_DIST_FORCE"):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                  # ONE JSON line, from rank 0
    return json.loads(lines[0])


def _plain_bench(gpus, env_extra, timeout=900):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env.update(env_extra)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--ensembles", "4", "--frames", "16", "--steps", "2",
           "--warmup", "1"]            # (default --legs auto: none of the post-timing legs run when world > 1)
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_bench_gpus_2_starts_two_ranks_by_itself_or_fails():
    """`python bench.py --gpus 2` with no launcher around it must start two ranks itself (a fresh torch.distributed.run
    child) and must never come back as a one-rank result.  On this GPU-less host both ranks stop at "needs a gfx950
    GPU": the parent relays the failure -- non-zero exit, no JSON line."""
    r = _plain_bench(2, {})
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "needs a gfx950 GPU" in r.stderr or "device(s) visible" in r.stderr, r.stderr[-2000:]


def test_bench_refuses_a_world_that_is_not_gpus():
    """--gpus 2 inside a one-rank environment (WORLD_SIZE=1) is an error, not a one-GPU measurement labelled n_gpus 1."""
    r = _plain_bench(2, {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_plain_bench_gpus_2_reports_two_ranks():
    """The form the driver's SCALE run may use: `python bench.py --gpus 2`, no torchrun.  Two ranks are started by
    bench.py itself (sharing this box's one GPU, so gloo carries the reductions; with >= 2 devices the default RCCL
    backend is taken) and the line says n_gpus 2 with one per_rank row each."""
    r = _plain_bench(2, {"DABGPU_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["world"] == 2 and len(j["per_rank"]) == 2
    assert "gloo" in j["collective_backend"] and "NOT RCCL" in j["collective_backend"]      # the line says what carried the reductions
    assert [p["rank"] for p in j["per_rank"]] == [0, 1] and all(p["frames_per_s"] > 0 for p in j["per_rank"])
    assert all(0 < p["roofline_frac"] < 1 and p["mover_same_geometry_ms"] > 0 for p in j["per_rank"])
    # every rank's row says what the allocator gave it and where its ceiling figure comes from
    for p in j["per_rank"]:
        bp = p["buffer_placement"]
        assert bp["method"] == "plain hipMalloc pair" and "too small" in bp["fallback_reason"]      # (this small shape)
        assert "dabgpu_mover_frames_dev" in p["mover_source"] and abs(p["kernel_over_mover"] - p["front_end_kernel_ms"] / p["mover_same_geometry_ms"]) < 1e-9
    assert j["cpu_baseline"] is None and j["legs"].startswith("skipped")                     # world > 1: the timed step only
    assert j["fic_bit_exact"] is True and j["msc_bit_exact"] is True
    assert abs(j["value"] * j["ms_per_step"] * 1e-3 - 128) < 1e-3


@pytest.mark.gpu
def test_plain_bench_gpus_2_without_two_devices_fails_under_rccl():
    """With the default backend (RCCL) two ranks need two devices: on the one-GPU box the run must fail loudly."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has two devices")
    r = _plain_bench(2, {})
    assert r.returncode != 0 and "device(s) visible" in r.stderr


@pytest.mark.gpu
def test_bench_two_ranks_execute_end_to_end():
    """The N>1 path of bench.py as the driver launches it -- torch.distributed.run, one process per rank, started as
    a fresh child before anything in it touches the GPU -- with two ranks sharing the one GPU of the test box (gloo
    carries the barrier and the three scalar reductions there; on a multi-GPU node the same code runs over RCCL).
    Both ranks shard the global ensemble list, decode their own streams bit-exactly, and rank 0 reports the sum."""
    one = _run_bench(1)
    two = _run_bench(2, {"DABGPU_DIST_BACKEND": "gloo"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    for r in (one, two):
        assert r["fic_bit_exact"] is True and r["msc_bit_exact"] is True and r["scaling"] == "weak"
        assert r["config"]["frames_per_step_per_gpu"] == 64
    # whole-job value: both ranks' frames over the slowest rank's time
    frames_two = two["value"] * two["ms_per_step"] * 1e-3
    frames_one = one["value"] * one["ms_per_step"] * 1e-3
    assert abs(frames_one - 64) < 1e-6 * 64 and abs(frames_two - 128) < 1e-6 * 128


@pytest.mark.gpu
def test_bench_config5_shape_512_ensembles_over_8_ranks():
    """BASELINE config 5's sharding -- 512 ensembles, 64 per rank, 8 ranks -- executed with the eight ranks sharing the
    test box's one GPU (16 frames per ensemble instead of 256 so that eight copies fit comfortably): every rank owns
    the global ids `id % 8 == rank`, decodes its 64 streams bit-exactly, and rank 0 reports the whole job."""
    import time
    t0 = time.time()
    r = _run_bench(8, {"DABGPU_DIST_BACKEND": "gloo"}, ensembles=64, frames=16)
    assert time.time() - t0 < 60.0                           # eight ranks sharing one GPU: the run stays boring
    assert r["n_gpus"] == 8 and len(r["per_rank"]) == 8 and r["fic_bit_exact"] is True and r["msc_bit_exact"] is True
    assert r["config"]["ensembles_per_gpu"] == 64 and r["config"]["frames_per_step_per_gpu"] == 1024
    assert abs(r["value"] * r["ms_per_step"] * 1e-3 - 8 * 1024) < 1e-3


@pytest.mark.gpu
def test_bench_rccl_branch_executes_with_one_rank():
    """The `nccl` (= RCCL) branch of bench.py -- process group on the GPU, barrier, the three all-reduces of the report
    on device tensors -- launched through torch.distributed.run with a single rank, which is all a one-GPU box can give
    RCCL (two ranks on one device are refused); the multi-rank logic is covered by the gloo runs above."""
    r = _run_bench(1, {"DABGPU_DIST_FORCE": "1", "DABGPU_DIST_BACKEND": "nccl"})
    assert r["n_gpus"] == 1 and r["fic_bit_exact"] is True and r["msc_bit_exact"] is True
    assert r["collective_backend"] == "nccl (RCCL)" and r["world"] == 1
    bp = r["config"]["buffer_placement"]             # (64 frames are far too small for the domains to matter: a plain pair, and it says why)
    assert bp["requested"] == "domains" and bp["method"] == "plain hipMalloc pair" and "too small" in bp["fallback_reason"]
    assert bp["setup_peak_over_final_footprint"] == 1.0
    assert abs(r["value"] * r["ms_per_step"] * 1e-3 - 64) < 1e-3


@pytest.mark.gpu
def test_bench_line_keeps_the_contract():
    """The one JSON line of a (small) default-shaped run carries every field the driver and the judge read, with the
    types and relations the contract states: whole-job value, roofline priced on the bytes the timed step moves, the CPU
    baseline of a bounded sample, both fine-frequency loops compared, nothing decoded wrongly."""
    env = dict(os.environ)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--ensembles", "4", "--frames", "16", "--steps", "3", "--warmup", "1",
           "--cpu-seconds", "2", "--sustained-seconds", "0.2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["metric"].startswith("DAB Mode-I frames/sec") and j["unit"] == "frames/s" and j["higher_is_better"] is True
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert j["dtype"] == "f32" and j["data"] == "synthetic" and "workload" in j["config"] and "model" not in j["config"]
    assert abs(j["value"] * j["ms_per_step"] * 1e-3 - 64) < 1e-6 * 64                       # frames per step / time per step
    assert j["fic_bit_exact"] is True and j["msc_bit_exact"] is True
    ro = j["roofline"]
    assert ro["bound"] == "hbm" and ro["unit"] == "GB/s" and ro["peak"] == 8000.0
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-9 and 0 < ro["frac"] < 1
    # the headline is the reference's data flow (every cyclic prefix read, the loop on the correlations): VERDICT r04 item 1
    assert ro["algorithmic_bytes_per_frame"] == 76 * 2552 * 8 + 230400 == 1782016 and ro["frames_per_launch"] == 64
    assert "cyclic-prefix" in j["config"]["estimator"] and "cyclic-prefix" in j["config"]["frequency_correction"]
    assert abs(ro["achieved"] - ro["algorithmic_bytes_per_frame"] * 64 / (ro["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * ro["achieved"]
    assert ro["avg_launch_ms"] < j["ms_per_step"]
    sm = j["step_ms"]                                          # spread of the timed steps (device time of each)
    assert 0 < sm["min"] <= sm["median"] <= sm["max"] and sm["min"] <= j["ms_per_step"] * 1.5
    assert abs(ro["box_mover_frac"] - ro["mover_same_geometry_GBps"] / 8000.0) < 1e-9 and 0 < ro["box_mover_frac"] < 1
    # HBM bytes per launch: measured in this very run by two rocprofv3 --pmc child processes (null only where the profiler
    # is not installed: the tracked file does not describe this launch size)
    if ro["traffic"] is not None:
        assert ro["traffic_source"].startswith("measured in this run")
        assert 0.98 < ro["traffic_over_algorithmic"] < 1.5 and abs(ro["traffic"] / (ro["algorithmic_bytes_per_frame"] * 64) - ro["traffic_over_algorithmic"]) < 1e-6
    # the stated ceiling: a mover of the kernel's own geometry on the timed buffers (the kernel cannot beat its own bytes)
    assert ro["mover_same_geometry_ms"] > 0 and "copy_ceiling" not in ro
    assert abs(ro["kernel_over_mover"] - ro["avg_launch_ms"] / ro["mover_same_geometry_ms"]) < 1e-9
    # (64 frames are decoded by the wave-per-codeword kernels: the lane decoder's record is null here and asserted by
    # test_bench_line_carries_the_decoder_roofline)
    assert j["decoder"]["kernels"] == "auto" and j["decoder"]["roofline"] is None and j["decoder"]["acs_per_s"] > 0
    bp = j["config"]["buffer_placement"]
    assert bp["requested"] == "domains" and bp["setup_peak_over_final_footprint"] <= 1.5   # the default; this small shape falls back to a plain pair
    assert bp["method"] == "plain hipMalloc pair" and "too small" in bp["fallback_reason"]
    # BASELINE configs 2 and 3 (one ensemble) and the plugin's one-frame-at-a-time use
    se = j["single_ensemble"]
    assert se["ofdm_fic"]["fic_bit_exact"] is True and se["ofdm_fic"]["value"] > 0 and se["ofdm_fic"]["frames_per_step"] == 16
    assert se["ofdm_fic_msc64"]["fic_bit_exact"] is True and se["ofdm_fic_msc64"]["msc_bit_exact"] is True
    for row in (se["ofdm_fic"], se["ofdm_fic_msc64"]):         # device time of the step's two calls
        assert 0 < row["front_end_call_ms"] and 0 < row["decode_call_ms"]
    hf = se["host_fed_per_frame"]
    assert hf["every_frame_locked"] is True and hf["fic_bit_exact"] is True and hf["msc_bit_exact"] is True
    assert hf["frame_ms"] > 0 and abs(hf["frame_ms"] - hf["ofdm_demod_stream_frame_ms"] - hf["decode_stream_frames_ms"]) < 1e-9
    # the host-fed figure BASELINE.md section 4 item 5 promises: synchronous calls and the ring
    hfed = j["host_fed"]
    assert hfed["frames_per_call"] == 16 and hfed["h2d_copy_GBps"] > 0
    assert hfed["synchronous_calls"]["fic_bit_exact"] is True and hfed["ring"]["fic_bit_exact"] is True
    assert hfed["ring"]["msc_bit_exact"] is True and hfed["ring"]["outputs_identical_to_synchronous_calls"] is True
    assert hfed["ring"]["value"] > 0 and hfed["ring_without_soft_bit_download"]["value"] > 0
    hm = j["host_mirror_end_to_end"]                          # the C++ mirror end to end: locked throughout, every FIB right, three services
    assert hm["frames"] >= 390 and hm["frames_desync"] == 0 and hm["fib_errors"] == 0 and hm["dab_plus_services_decoded"] == 3
    assert hm["all_superframes_clean"] is True and hm["value"] > 0
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "frames/s" and cb["value"] > 0 and cb["cores"] >= 1 and "frames" in cb["sample"]
    assert cb["simd_port"]["kind"] == "simd_port" and cb["simd_port"]["decodes_bench_inputs_to_transmitted_fibs"] is True
    # the library's own estimator: a named leg, and copied to the top level beside `value`
    dd = j["with_decision_directed_loop"]
    assert dd["outputs_identical_to_timed_run"] is True and dd["max_abs_soft_bit_difference"] <= 1
    assert dd["algorithmic_bytes_per_frame"] == (76 * 2048 + 504) * 8 + 230400
    assert j["value_own_estimator"] == dd["value"] and ro["frac_own_estimator"] == dd["roofline_frac"]
    assert se["ofdm_fic_own_estimator"]["fic_bit_exact"] is True and se["ofdm_fic_msc64_own_estimator"]["msc_bit_exact"] is True
    assert j["sustained"]["outputs_identical_to_timed_run"] is True and j["sustained"]["steps"] >= 1
    assert j["selective_soft_output"]["outputs_identical_to_whole_frame_run"] is True
    cl = j["closed_loop"]
    assert cl["fic_bit_exact"] is True and cl["msc_bit_exact"] is True and cl["tracking"]["fic_bit_exact"] is True
    assert cl["tracking"]["streams_tracking"] == 4 and cl["tracking"]["frames_desync_total"] == 0


def test_self_launch_parent_never_loads_torch():
    """`python bench.py --gpus N` without a launcher: the parent starts the N ranks as a fresh `torch.distributed.run` child
    and relays its line.  It must reach that point -- and end -- without torch (let alone torch.cuda) in the process: a
    parent that had initialised the GPU runtime would hand its state to every rank (and on this pool an exec / fork from
    such a process takes the machine down).  The child is replaced by a stand-in that prints one line."""
    code = r"""
import json, subprocess, sys
sys.argv = ['bench.py', '--gpus', '2', '--steps', '1']
sys.path.insert(0, %r)
import bench
seen = {}
class FakePopen:
    def __init__(self, cmd, **kw):
        seen['cmd'] = cmd; seen['env'] = kw.get('env', {})
        self.returncode = 0
    def communicate(self):
        return json.dumps({'n_gpus': 2, 'value': 1.0}) + '\n', None
subprocess.Popen = FakePopen
try:
    bench.main()
    raise AssertionError('self_launch returned')
except SystemExit as e:
    assert e.code == 0, e.code
assert 'torch' not in sys.modules and 'torch.cuda' not in sys.modules, 'the launcher process imported torch'
assert 'dabgpu' not in sys.modules
cmd = seen['cmd']
assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '2'
assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and seen['env'].get('HSA_ENABLE_IPC_MODE_LEGACY') == '0'
print('ok')
""" % ROOT
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout + r.stderr)[-2000:]


@pytest.mark.gpu
def test_bench_line_carries_the_decoder_roofline():
    """the 19 % of the step the channel decoder takes has a tracked fraction.  `decoder.roofline` is built
    from the library's own HIP-event timers around the lane decoder's kernels during the timed steps (forward pass | traceback |
    history copy) and, when rocprofv3 is there, from the TCC counters of the same two child passes that give roofline.traffic.
    Run at a small shape with the lane kernels pinned (--decoder lane; the default shape takes them by itself) and every leg
    but the traffic measurement switched off."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--ensembles", "4", "--frames", "16", "--steps", "3", "--warmup", "1",
           "--decoder", "lane", "--legs", "all", "--cpu-seconds", "0", "--sustained-seconds", "0", "--no-dd-leg", "--no-fft-stage",
           "--no-selective", "--no-closed-loop", "--no-single-ensemble", "--no-host-fed", "--no-host-mirror"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ), cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["fic_bit_exact"] is True and j["msc_bit_exact"] is True and j["decoder"]["kernels"] == "lane"
    d = j["decoder"]["roofline"]
    assert d["bound"] == "valu" and d["valu_per_step"] == 202 and d["acs_valu_per_step"] == 96 and "lane_forward_grouped_kernel" in d["kernel"]
    assert d["frames_per_launch"] == 64 and d["wave_steps_per_launch"] == 4 * (774 + 1542) and d["launches_timed"] == 3
    assert d["algorithmic_bytes_per_frame"] == 9216 + 396 + 37632 == 47244          # A_fic + A_msc64
    assert d["survivor_bytes_per_launch"] == 256 * (774 + 1542) * 8
    bound = d["wave_steps_per_launch"] * 96 * d["cycles_per_packed_int16_instruction"] / (d["simds"] * d["clock_GHz"] * 1e9) * 1e3
    assert abs(d["acs_only_bound_ms"] - bound) < 1e-12 and abs(d["valu_issue_bound_ms"] - bound * 202 / 96) < 1e-12
    assert abs(d["frac_of_acs_bound"] - d["acs_only_bound_ms"] / d["forward_ms"]) < 1e-12 and 0 < d["frac_of_acs_bound"] < d["frac_of_valu_issue_bound"] < 1
    # the three parts are what the decode call's device time is made of (the events between them cost a few microseconds)
    assert 0 < d["history_ms"] < d["traceback_ms"] < d["forward_ms"]
    parts = d["forward_ms"] + d["traceback_ms"] + d["history_ms"]
    assert parts <= j["decoder"]["fic_and_msc_ms"] * 1.02 and parts >= 0.5 * j["decoder"]["fic_and_msc_ms"]
    assert abs(d["traceback_GBps_on_survivor_bytes"] - d["survivor_bytes_per_launch"] / (d["traceback_ms"] * 1e-3) / 1e9) < 1e-6
    if j["roofline"]["traffic"] is not None and j["roofline"]["traffic_source"].startswith("measured in this run"):
        assert d["traffic_source"].startswith("measured in this run")
        tk = d["traffic_by_kernel"]
        assert set(tk) == {"lane_forward_grouped_kernel", "lane_traceback_grouped_kernel", "msc_history_kernel"}
        assert abs(d["traffic"] - sum(v["read"] + v["write"] for v in tk.values())) < 1.0
        assert abs(d["traffic_over_algorithmic"] - d["traffic"] / (47244 * 64)) < 1e-9
        # (no physical lower bound is asserted at this toy shape: 4.7 MB of survivors may never leave the L2; at the default
        # shape the forward pass writes and the traceback reads 1.21 GB each, profiles/r06_pmc_traffic_decoder.json)
        assert tk["lane_forward_grouped_kernel"]["write"] > 0 and tk["lane_traceback_grouped_kernel"]["read"] > 0
        assert d["traffic_over_algorithmic"] > 0 and d["traceback_GBps"] > 0
