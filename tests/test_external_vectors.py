"""One-command pin of the oracle and the HIP path against vectors dumped from an upstream build.

`tests/external/*.npz` (absent in this repository: nothing upstream can be built here) are run through the CPU oracle
and -- with `-m gpu` -- through libdabgpu with the tolerances of DESIGN.md section 2 (FIB / MSC bytes and CRC flags
bit-exact, soft bits within 1 LSB; a pure sign or scale convention difference of the soft bits is named as such).
What to dump and where: INTEGRATION.md section 6.  The committed self-generated sample keeps the harness exercised:
green on it, RED on a deliberately sign-flipped copy."""
import glob
import os

import numpy as np
import pytest

from conftest import ROOT, golden_path
from external_vectors import check_file, soft_convention

EXTERNAL = sorted(glob.glob(os.path.join(ROOT, "tests", "external", "*.npz")))
SAMPLE = golden_path("external_sample.npz")


def flipped_copy(tmp_path, what):
    d = dict(np.load(SAMPLE))
    if what == "soft":                       # the dump's soft bits with the opposite sign convention
        d["soft"] = (-d["soft"].astype(np.int16)).astype(np.int8)
    elif what == "fib":                      # one flipped bit in one FIB
        d["fib"] = d["fib"].copy(); d["fib"][2, 5, 7] ^= 0x10
    elif what == "msc":
        d["msc_7"] = d["msc_7"].copy(); d["msc_7"][17, 3] ^= 1
    elif what == "crc":
        d["crc_ok"] = d["crc_ok"].copy(); d["crc_ok"][1, 4] = 0
    path = tmp_path / ("flipped_%s.npz" % what)
    np.savez_compressed(path, **d)
    return path


def test_sample_file_is_green_through_the_oracle(built):
    r = check_file(SAMPLE, "oracle")
    assert r["soft"] == ("equal", 0) and r["fib_crc_flags_equal"] and r["fibs_compared"] == 60 and r["msc_7"] == 5


@pytest.mark.parametrize("what,msg", [("soft", "'sign'"), ("fib", "FIB bytes differ"), ("msc", "MSC bytes"), ("crc", "CRC flags differ")])
def test_flipped_copies_are_red_through_the_oracle(built, tmp_path, what, msg):
    with pytest.raises(AssertionError) as e:
        check_file(flipped_copy(tmp_path, what), "oracle")
    assert msg in str(e.value) or (what == "soft" and "sign" in str(e.value)), str(e.value)


def test_convention_report():
    rng = np.random.default_rng(1)
    a = rng.integers(-127, 128, 5000).astype(np.int8)
    assert soft_convention(a, a) == ("equal", 0)
    assert soft_convention(a, (-a.astype(np.int16)).astype(np.int8))[0] == "sign"
    kind, factor = soft_convention(a, (a.astype(np.int16) // 2).astype(np.int8))
    assert kind == "scale" and abs(factor - 0.5) < 0.05
    assert soft_convention(a, rng.integers(-127, 128, 5000).astype(np.int8))[0] == "different"


@pytest.mark.skipif(not EXTERNAL, reason="no upstream dumps under tests/external (see tests/external/README.md)")
@pytest.mark.parametrize("path", EXTERNAL)
def test_external_vectors_through_the_oracle(built, path):
    print(check_file(path, "oracle"))


# ------------------------------------------------------------------ the HIP path
@pytest.mark.gpu
def test_sample_file_is_green_through_the_hip_path(ctx):
    r = check_file(SAMPLE, ctx)
    assert r["soft"][0] == "equal" and r["soft"][1] <= 1 and r["fib_crc_flags_equal"] and r["msc_7"] == 5


@pytest.mark.gpu
@pytest.mark.parametrize("what", ["soft", "fib", "msc", "crc"])
def test_flipped_copies_are_red_through_the_hip_path(ctx, tmp_path, what):
    with pytest.raises(AssertionError):
        check_file(flipped_copy(tmp_path, what), ctx)


@pytest.mark.gpu
@pytest.mark.skipif(not EXTERNAL, reason="no upstream dumps under tests/external (see tests/external/README.md)")
@pytest.mark.parametrize("path", EXTERNAL)
def test_external_vectors_through_the_hip_path(ctx, path):
    print(check_file(path, ctx))
