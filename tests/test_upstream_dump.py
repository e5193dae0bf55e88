"""tools/upstream_dump: the header-only dumper a maintainer drops into an upstream build (radio_block.cpp:25 / :42) and
its converter to the tests/external/*.npz format.  It cannot meet upstream here; what CAN be held:
it compiles, warning-free, against the types the plugin uses at those two points (through the host mirror's headers), and
a replay of tests/golden/external_sample.npz through it -- from two threads, the sub-channel starting mid-frame as a
16-CIF de-interleaver does -- converts back to exactly that file's arrays, which the external-vector harness accepts."""
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT, golden_path

TOOL = os.path.join(ROOT, "tools", "upstream_dump")
HOST = os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd", "host")
sys.path.insert(0, TOOL)


def test_dumper_compiles_against_the_mirror_headers():
    cmd = ["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I" + TOOL, "-I" + HOST,
           "-I" + os.path.join(ROOT, "include"), os.path.join(TOOL, "wire_mirror_check.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_replay_of_the_sample_round_trips_through_dumper_and_converter(tmp_path, built):
    import to_npz
    import external_vectors as X
    d = np.load(golden_path("external_sample.npz"))
    n = d["soft"].shape[0]
    ident = "7"
    desc = [int(v) for v in d["subchannel_" + ident]]
    msc = d["msc_" + ident]
    first = int(d["msc_valid_from"])
    pin = str(tmp_path / "in")
    d["soft"].tofile(pin + ".soft.bin"); d["fib"].tofile(pin + ".fib.bin"); d["crc_ok"].astype(np.uint8).tofile(pin + ".crc.bin")
    msc.tofile(pin + ".msc.bin"); d["iq"].tofile(pin + ".iq.bin"); d["freq_offset"].tofile(pin + ".fo.bin")
    exe = str(tmp_path / "replay")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-pthread", "-I" + TOOL,
                           os.path.join(TOOL, "replay.cpp"), "-o", exe])
    pout = str(tmp_path / "dump")
    r = subprocess.run([exe, pin, pout, str(n), ident, str(msc.shape[1]), str(first)] + [str(v) for v in desc])
    assert r.returncode == 0
    out = str(tmp_path / "back.npz")
    arrays = to_npz.convert(pout, out)
    back = np.load(out)
    assert set(back.files) == set(d.files)
    for k in d.files:
        a, b = np.asarray(d[k]), np.asarray(back[k])
        if k == "msc_" + ident:
            a, b = a[first:], b[first:]                     # (rows before msc_valid_from are "ignored" by the format)
        assert a.shape == b.shape and (a == b).all(), k
    assert "soft_frames %d" % n in open(pout + ".meta.txt").read()
    # ... and the harness takes the converted file as it takes the sample
    rep = X.check_file(out, "oracle")
    assert rep["fib_crc_flags_equal"] and rep["msc_" + ident] == 4 * n - first
    assert arrays["msc_valid_from"] == first
