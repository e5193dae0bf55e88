"""the host mirror -- OFDM_Demod, the ring, BasicRadio, the DAB+ channel code -- runs on two threads
beside a polling GUI thread and only the FIG parser had ever seen a sanitizer.  GPU sanitizers are not
available on the pool, so the C ABI under the mirror is replaced by a TEST-ONLY fake that calls the oracle
(tests/fake_abi/fake_dabgpu.cpp: never shipped, never on the product path) and the plugin's wiring -- the reference's own
src/radio_block.cpp when it is mounted, a local equivalent otherwise -- is run on a synthetic multiplex under
ThreadSanitizer and under AddressSanitizer + UBSan while a third thread polls every getter, moves the configuration's
knobs and presses both "Reset" buttons.  Both must come out clean, and the stream must still decode.
The ASan build links oracle/liboracle_asan.so (`make -C oracle liboracle_asan.so`): the oracle's C code runs sanitised too."""
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
SERVICES = [("Radio One", 0xC221, 3, 0, 3, 64, 0), ("Jazz 24", 0xC222, 7, 0, 2, 48, 48)]
REF = "/root/reference/src/radio_block.cpp"


@pytest.fixture(scope="module")
def stream(tmp_path_factory):
    """13 frames of a two-service DAB+ multiplex through a mild channel, with a lead-in for the level estimator"""
    ens = synth.ServiceEnsemble(1, SERVICES, n_frames=5)
    iq = np.tile(ens.iq().ravel(), 3)[:13 * synth.NB_FRAME_SAMPLES]
    iq = synth.channel(iq, snr_db=18.0, cfo=0.7 / 2048, rng=np.random.default_rng(11))
    path = tmp_path_factory.mktemp("iq") / "iq.cf32"
    np.concatenate([iq[-30000:], iq, iq[:synth.NB_NULL + 5000]]).astype(np.complex64).tofile(path)
    return str(path)


def build(tmp_path, sanitizer, use_reference):
    exe = str(tmp_path / ("mirror_threads_" + sanitizer.split(",")[0]))
    inc = ["-I" + HOST, "-I" + os.path.join(ROOT, "include"), "-I" + ORACLE, "-I" + FAKE]
    srcs = [os.path.join(FAKE, "mirror_threads.cpp"), os.path.join(FAKE, "fake_dabgpu.cpp")] + [os.path.join(HOST, s) for s in MIRROR_SRC]
    flags = ["-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-pthread", "-fsanitize=" + sanitizer, "-fno-sanitize-recover=all"]
    if use_reference:
        flags.append("-DUSE_REFERENCE_RADIO_BLOCK")
        inc.append("-I/root/reference/src")
        srcs.append(REF)                                           # compiled from where it lies; nothing is copied
    if "address" in sanitizer:
        subprocess.check_call(["make", "-C", ORACLE, "liboracle_asan.so"], stdout=subprocess.DEVNULL)
        link = ["-L" + ORACLE, "-loracle_asan", "-Wl,-rpath," + ORACLE]
    else:
        # ThreadSanitizer wants every object instrumented: the oracle's C sources go into the binary
        objs = []
        for c in ("dab_oracle.c", "dabplus_oracle.c"):
            o = str(tmp_path / (c + "." + sanitizer.split(",")[0] + ".o"))
            subprocess.check_call(["gcc", "-O1", "-g", "-std=c99", "-ffp-contract=off", "-fsanitize=" + sanitizer, "-c", os.path.join(ORACLE, c), "-o", o])
            objs.append(o)
        link = objs + ["-lm"]
    subprocess.check_call(["g++"] + flags + inc + srcs + link + ["-o", exe])
    return exe


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_mirror_threads_run_clean_under_sanitizers(tmp_path, stream, sanitizer):
    exe = build(tmp_path, sanitizer, os.path.exists(REF))
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1 suppressions=" + os.path.join(FAKE, "tsan.supp"), ASAN_OPTIONS="detect_leaks=1:abort_on_error=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([exe, stream, "10007"], capture_output=True, text=True, timeout=900, env=env)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), out[-6000:]
    assert "WARNING: ThreadSanitizer" not in out and "ERROR: AddressSanitizer" not in out and "runtime error" not in out, out[-6000:]
    line = dict(kv.split("=") for kv in r.stdout.splitlines()[-2].split())
    head = dict(kv.split("=") for kv in r.stdout.splitlines()[-3].split())
    # the stream was really processed: frames before AND after the demodulator's reset, FIBs decoded by the radio that
    # replaced the first one, audio channels opened from the database, both resets pressed, the GUI thread polling throughout
    assert int(head["frames_before_reset"]) >= 4 and int(head["frames_read_after_reset"]) >= 3 and int(head["state"]) == 4
    assert int(line["good_fibs_last_radio"]) >= 24 and int(line["channels_seen"]) == 2 and int(line["resets"]) == 2
    assert int(line["polls"]) > 20


@pytest.mark.skipif(not os.path.exists(REF), reason="reference not mounted")
def test_local_wiring_builds_too(tmp_path):
    """the stand-in for radio_block.cpp (machines without /root/reference) at least compiles and links"""
    build(tmp_path, "address,undefined", False)
