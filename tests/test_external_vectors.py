"""One-command pin of the oracle and the HIP path against vectors dumped from an upstream build.

`tests/external/*.npz` (absent in this repository: nothing upstream can be built here) are run through the CPU oracle
and -- with `-m gpu` -- through libdabgpu with the tolerances of DESIGN.md section 2 (FIB / MSC bytes and CRC flags
bit-exact, soft bits within 1 LSB; a pure sign or scale convention difference of the soft bits is named as such).
What to dump and where: INTEGRATION.md section 6.  The committed self-generated sample keeps the harness exercised:
green on it, RED on a deliberately sign-flipped copy.

PARITY IS UNPINNED while tests/external/ is empty.  The two `test_external_vectors_through_*` tests then run the harness on
tests/golden/external_sample.npz instead of skipping -- test id `self_generated_sample_pins_nothing`: the file was made by
this repository's own transmitter and oracle (tests/golden/make_external_sample.py), so a green run proves that the branch
a real dump will take executes (it had never run on the GPU) and says NOTHING about the reference.
The hand-over points a dump taps: /root/reference/src/radio_block.cpp:25 (On_OFDM_Frame payload) and :42 (BasicRadio::Process)."""
import glob
import os

import numpy as np
import pytest

from conftest import ROOT, golden_path
from external_vectors import check_file, soft_convention

EXTERNAL = sorted(glob.glob(os.path.join(ROOT, "tests", "external", "*.npz")))
SAMPLE = golden_path("external_sample.npz")
# what the two harness tests run on: real dumps when there are any, else the self-generated sample (which pins nothing)
VECTORS = EXTERNAL or [SAMPLE]


def _vector_id(path):
    return os.path.basename(path) if EXTERNAL else "self_generated_sample_pins_nothing"


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


@pytest.mark.parametrize("path", VECTORS, ids=_vector_id)
def test_external_vectors_through_the_oracle(built, path):
    print(("UPSTREAM DUMP: " if EXTERNAL else "self-generated sample, pins nothing: ") + str(check_file(path, "oracle")))


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
@pytest.mark.parametrize("path", VECTORS, ids=_vector_id)
def test_external_vectors_through_the_hip_path(ctx, path):
    print(("UPSTREAM DUMP: " if EXTERNAL else "self-generated sample, pins nothing: ") + str(check_file(path, ctx)))
