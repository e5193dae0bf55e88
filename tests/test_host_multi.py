"""Operating shapes of the boundary the plugin allows and no test had run:

(a) several plugin instances at once -- /root/reference/src/main.cpp:12 (MAX_INSTANCES -1), :22-30: every instance is a
    DABModule with a Radio_Block of its own, i.e. two libdabgpu contexts and two threads (src/radio_block.cpp:22-28, 33-44)
    that end every frame call on a watched word.  host/demo/dab_host_multi.cpp builds N such wirings in ONE process on ONE
    device and runs them together on different ensembles: every FIB and MSC byte right, no cross-talk between the
    instances' counters, per-frame time within 1.5 x the single-instance figure;
(b) a device call that fails under the mirror -- dabgpu_test_fail_frame_call makes the n-th frame call of one context
    return DABGPU_ERR_HIP: OFDM_Demod::Process / BasicRadio::Process do not throw (radio_block.cpp:27, 37: nobody could
    catch), the demodulator counts one desync and locks again on the next frame, the radio counts one lost frame, drops
    its de-interleaver state and keeps the FIC flowing (host/ofdm/ofdm_demodulator.cpp, host/basic_radio/basic_radio.cpp).

`-m "not gpu"`: (b) with the TEST-ONLY fake ABI of tests/fake_abi (the oracle behind the mirror; never shipped) -- the
mirror's branches.  `-m gpu`: (a) and (b) on libdabgpu."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from dabgpu import synth

HOST = os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd", "host")
FAKE = os.path.join(ROOT, "tests", "fake_abi")
ORACLE = os.path.join(ROOT, "oracle")
MIRROR_SRC = ["ofdm/ofdm_demodulator.cpp", "basic_radio/basic_radio.cpp", "basic_radio/basic_dab_plus_channel.cpp", "dab/fic/fic_parser.cpp"]


def make_stream(path, seed, n_frames, cfo):
    ens = synth.Ensemble(seed=seed, n_frames=n_frames)
    iq = synth.channel(ens.iq().ravel(), snr_db=18.0, cfo=cfo, rng=np.random.default_rng(seed))
    np.concatenate([iq[-30000:], iq, iq[:synth.NB_NULL + 5000]]).astype(np.complex64).tofile(path)
    return ens


def run_multi(exe, tmp_path, tag, files, env_extra=None, chunk=40961):
    prefix = str(tmp_path / tag)
    env = dict(os.environ)
    for k in ("DAB_MULTI_FAIL_OFDM", "DAB_MULTI_FAIL_RADIO"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([exe, prefix, str(chunk), "64", "0", "3"] + [str(f) for f in files], capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = [dict(kv.split("=") for kv in l.split()) for l in r.stdout.splitlines() if l.startswith("instance=")]
    assert len(rows) == len(files) and all(row["threw"] == "0" for row in rows), r.stdout
    out = []
    for i, row in enumerate(rows):
        p = "%s.%d" % (prefix, i)
        out.append((row, np.fromfile(p + ".fib", np.uint8).reshape(-1, 12, 32), np.fromfile(p + ".crc", np.uint8).reshape(-1, 12),
                    np.fromfile(p + ".msc", np.uint8).reshape(-1, 192)))
    total = float([l for l in r.stdout.splitlines() if l.startswith("instances=")][0].split("processing_s=")[1])
    return out, total


def check_clean_instance(row, fib, crc, msc, ens, n_frames):
    assert int(row["frames_read"]) == n_frames and int(row["frames_desync"]) == 0 and int(row["frames_lost"]) == 0, row
    assert int(row["frames_out"]) == n_frames and int(row["fibs"]) == 12 * n_frames and int(row["state"]) == 4, row
    assert fib.shape[0] == n_frames and crc[2:].all()
    for f in range(n_frames):
        if crc[f].all():
            assert (fib[f] == ens.fibs[f]).all(), f
    assert msc.shape[0] == 4 * n_frames - 15
    assert all((msc[i] == ens.msc_bytes[i]).all() for i in range(8, msc.shape[0]))


def check_ofdm_failure(row, fib, crc, msc, ens, n_frames, failed_call):
    """the demodulator's `failed_call`-th frame call (1-based: frame index failed_call - 1) reported a device failure"""
    lost = failed_call - 1
    assert int(row["frames_desync"]) == 1 and int(row["state"]) == 4, row                  # counted once, locked again
    # lock re-acquired within two frames: at most one more frame is missing beside the failed one
    assert n_frames - 2 <= fib.shape[0] <= n_frames - 1 and int(row["frames_out"]) == fib.shape[0], (row, fib.shape)
    sent = [f for f in range(n_frames) if f != lost]
    if fib.shape[0] == n_frames - 2:
        sent.remove(lost + 1)
    for k, f in enumerate(sent):
        if k >= 2:
            assert crc[k].all(), (k, f)
        if crc[k].all():
            assert (fib[k] == ens.fibs[f]).all(), (k, f)      # nothing out of order, nothing repeated
    assert int(row["frames_lost"]) == 0                          # the radio saw no failure of its own


def check_radio_failure(row, fib, crc, msc, ens, n_frames, failed_call):
    lost = failed_call - 1
    assert int(row["frames_lost"]) == 1 and int(row["frames_desync"]) == 0 and int(row["frames_read"]) == n_frames, row
    # the FIC keeps flowing: the failed frame's FIBs come from the FIC-only call (basic_radio.cpp:84-95)
    assert fib.shape[0] == n_frames and int(row["fibs"]) == 12 * n_frames
    for f in range(n_frames):
        if f >= 2:
            assert crc[f].all(), f
        if crc[f].all():
            assert (fib[f] == ens.fibs[f]).all(), f
    # the sub-channel starts over: logical frames of the frames before the failure, then 16 CIFs of warm-up after it
    before = max(0, 4 * lost - 15)
    after = max(0, 4 * (n_frames - lost - 1) - 15)
    assert msc.shape[0] == before + after, (msc.shape, before, after)
    for i in range(8, before):
        assert (msc[i] == ens.msc_bytes[i]).all(), i
    for j in range(after):
        assert (msc[before + j] == ens.msc_bytes[4 * (lost + 1) + j]).all(), j


# --------------------------------------------------------------------------------------------- CPU: the mirror's branches
@pytest.fixture(scope="module")
def fake_multi(tmp_path_factory, built):
    """dab_host_multi over the TEST-ONLY fake ABI (the oracle's C code behind the mirror)"""
    d = tmp_path_factory.mktemp("fake_multi")
    exe = str(d / "dab_host_multi_fake")
    inc = ["-I" + HOST, "-I" + os.path.join(ROOT, "include"), "-I" + ORACLE, "-I" + FAKE]
    srcs = [os.path.join(HOST, "demo", "dab_host_multi.cpp"), os.path.join(FAKE, "fake_dabgpu.cpp")] + [os.path.join(HOST, s) for s in MIRROR_SRC]
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread"] + inc + srcs + ["-L" + ORACLE, "-loracle", "-Wl,-rpath," + ORACLE, "-o", exe])
    return exe


def test_mirror_survives_a_failing_frame_call_fake_abi(fake_multi, tmp_path):
    n = 12
    ens = make_stream(tmp_path / "a.cf32", 301, n, 0.21 / 2048)
    (row, fib, crc, msc), = run_multi(fake_multi, tmp_path, "clean", [tmp_path / "a.cf32"])[0]
    check_clean_instance(row, fib, crc, msc, ens, n)
    (row, fib, crc, msc), = run_multi(fake_multi, tmp_path, "ofdm", [tmp_path / "a.cf32"], {"DAB_MULTI_FAIL_OFDM": "0:6"})[0]
    check_ofdm_failure(row, fib, crc, msc, ens, n, 6)


def test_radio_survives_a_failing_decode_call_fake_abi(fake_multi, tmp_path):
    n = 12
    ens = make_stream(tmp_path / "a.cf32", 302, n, -0.12 / 2048)
    (row, fib, crc, msc), = run_multi(fake_multi, tmp_path, "radio", [tmp_path / "a.cf32"], {"DAB_MULTI_FAIL_RADIO": "0:6"})[0]
    check_radio_failure(row, fib, crc, msc, ens, n, 6)


# --------------------------------------------------------------------------------------------- GPU: libdabgpu
@pytest.fixture(scope="module")
def multi(built):
    exe = os.path.join(HOST, "dab_host_multi")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", HOST, "-j4"], stdout=subprocess.DEVNULL)
    return exe


@pytest.mark.gpu
def test_two_radio_blocks_at_once_on_one_device(multi, tmp_path):
    """Four contexts, four threads (+ the two feeders' work inside OFDM_Demod::Process), two different ensembles at two
    different carrier offsets: both instances decode their own multiplex completely, their counters do not mix, and an
    instance's time per frame stays within 1.5 x what it is alone."""
    n = 40
    ens = [make_stream(tmp_path / "a.cf32", 311, n, 0.23 / 2048), make_stream(tmp_path / "b.cf32", 312, n, -1.31 / 2048)]
    assert not (ens[0].fibs[0] == ens[1].fibs[0]).all()                      # two different multiplexes
    files = [tmp_path / "a.cf32", tmp_path / "b.cf32"]

    def per_frame_ms(rows):
        return [1e3 * (float(r[0]["ofdm_process_s"]) + float(r[0]["radio_process_s"])) / n for r in rows]

    best = None
    for attempt in range(3):                                                 # (timing on a shared box: the best of three)
        alone = [run_multi(multi, tmp_path, "alone%d" % i, [files[i]])[0] for i in range(2)]
        both, wall = run_multi(multi, tmp_path, "both", files)
        for i in range(2):
            check_clean_instance(*alone[i][0], ens[i], n)
            check_clean_instance(*both[i], ens[i], n)
            # the same bytes whether the instance runs alone or beside another one
            assert (both[i][1] == alone[i][0][1]).all() and (both[i][3] == alone[i][0][3]).all()
        t_alone = [per_frame_ms(alone[i])[0] for i in range(2)]
        t_both = per_frame_ms(both)
        ratio = max(t_both[i] / t_alone[i] for i in range(2))
        best = ratio if best is None else min(best, ratio)
        print("attempt %d: per-frame ms alone %s, together %s, ratio %.2f, wall %.3f s" % (attempt, t_alone, t_both, ratio, wall))
        if best <= 1.5:
            break
    assert best <= 1.5, best
    # ... and both ran faster than real time together (96 ms per frame each)
    assert max(t_both) < 96.0


@pytest.mark.gpu
def test_mirror_survives_a_failing_frame_call_on_the_device(multi, tmp_path):
    """instance 0's sixth frame call reports DABGPU_ERR_HIP while instance 1 runs beside it untouched"""
    n = 12
    ens = [make_stream(tmp_path / "a.cf32", 321, n, 0.4 / 2048), make_stream(tmp_path / "b.cf32", 322, n, 0.0)]
    rows, _ = run_multi(multi, tmp_path, "ofdm", [tmp_path / "a.cf32", tmp_path / "b.cf32"], {"DAB_MULTI_FAIL_OFDM": "0:6"})
    check_ofdm_failure(*rows[0], ens[0], n, 6)
    check_clean_instance(*rows[1], ens[1], n)


@pytest.mark.gpu
def test_radio_survives_a_failing_decode_call_on_the_device(multi, tmp_path):
    n = 12
    ens = [make_stream(tmp_path / "a.cf32", 331, n, 0.0), make_stream(tmp_path / "b.cf32", 332, n, -0.3 / 2048)]
    rows, _ = run_multi(multi, tmp_path, "radio", [tmp_path / "a.cf32", tmp_path / "b.cf32"], {"DAB_MULTI_FAIL_RADIO": "1:6"})
    check_clean_instance(*rows[0], ens[0], n)
    check_radio_failure(*rows[1], ens[1], n, 6)
