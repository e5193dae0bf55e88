"""Device code of the fused OFDM front end, read from the compiler's own metadata (`hipcc -S --cuda-device-only`, the
flags of csrc/Makefile; cross-compiles without a GPU): none of the eight instantiations of ofdm_wave_kernel spills a
register or uses scratch memory, and all keep the occupancy DESIGN.md 4.1 states (<= 168 VGPRs: three 4-wave workgroups
per CU).  VERDICT r05 item 3: the instantiation the plugin runs while its GUI polls GetFrameDataVec() --
<FFT_ONLY=false, WITH_DQPSK=true, SELECT=false, NCO=true>, host/ofdm/ofdm_demodulator.cpp, the consumer being
/root/reference/src/render_radio_block.cpp:109, 887-918 -- had 3 VGPR spills / 16 B of scratch."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd", "csrc")


def kernel_metadata(asm):
    """{kernel name: {field: int}} from the amdhsa.kernels list of a device assembly listing (or of `llvm-readelf --notes`):
    one `- .field:` item per kernel, fields in alphabetical order (so .name comes after .group_segment_fixed_size)."""
    out, cur, indent = {}, None, None
    for line in asm.splitlines():
        m = re.match(r"(\s*)- \.(\w+):\s*(\S*)", line)
        if m and (indent is None or len(m.group(1)) == indent) and m.group(2) in ("agpr_count", "args"):
            indent = len(m.group(1))
            cur = {}
            line = m.group(1) + "  ." + m.group(2) + ": " + m.group(3)
        if cur is None:
            continue
        m = re.match(r"\s*\.name:\s+(\S+)", line)
        if m and len(line) - len(line.lstrip()) == indent + 2:
            out[m.group(1)] = cur
            continue
        m = re.match(r"\s*\.(vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|vgpr_count|agpr_count|group_segment_fixed_size):\s+(\d+)", line)
        if m:
            cur[m.group(1)] = int(m.group(2))
    return out


@pytest.fixture(scope="module")
def ofdm_asm(tmp_path_factory):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.fail("hipcc not found: the device assembly cannot be checked")
    mk = open(os.path.join(CSRC, "Makefile")).read()
    flags = re.search(r"^CXXFLAGS\s*\?=\s*(.*)$", mk, re.M).group(1).replace("$(ARCH)", "gfx950").replace("$(EXTRA)", "").split()
    out = tmp_path_factory.mktemp("asm") / "ofdm_kernels.s"
    subprocess.check_call([hipcc] + flags + ["-S", "--cuda-device-only", os.path.join(CSRC, "ofdm_kernels.hip"), "-o", str(out)],
                          stderr=subprocess.DEVNULL)
    return out.read_text()


def test_no_front_end_instantiation_spills(ofdm_asm):
    md = {k: v for k, v in kernel_metadata(ofdm_asm).items() if "ofdm_wave_kernel" in k}
    # <FFT_ONLY, WITH_DQPSK, SELECT, NCO>: the fused kernel x {plain, DQPSK out, selection} x {NCO, none} + the FFT stage x {NCO, none}
    want = {"ILb0ELb0ELb0ELb1E", "ILb0ELb0ELb0ELb0E", "ILb0ELb1ELb0ELb1E", "ILb0ELb1ELb0ELb0E", "ILb0ELb0ELb1ELb1E", "ILb0ELb0ELb1ELb0E",
            "ILb1ELb0ELb0ELb1E", "ILb1ELb0ELb0ELb0E"}
    got = {re.search(r"ofdm_wave_kernel(ILb\dELb\dELb\dELb\dE)", k).group(1) for k in md}
    assert got == want, got ^ want
    for k, v in md.items():
        assert v["vgpr_spill_count"] == 0 and v["sgpr_spill_count"] == 0 and v["private_segment_fixed_size"] == 0, (k, v)
        assert v["vgpr_count"] <= 168 and v.get("agpr_count", 0) == 0, (k, v)      # 512 / 3 workgroups of 4 waves per SIMD set
        assert v["group_segment_fixed_size"] <= 160 * 1024 // 3, (k, v)              # 3 workgroups per CU share 160 KB of LDS


def test_the_scratch_free_claim_holds_for_the_decoders_too():
    """The Viterbi kernels (wave and lane) and the DAB+ kernels use no scratch memory either; what the listing says is
    compared against the compiled object the library is linked from when it is there."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdump = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    lib = os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd", "libdabgpu.so")
    if not (os.path.exists(lib) and os.path.exists(objdump) and os.path.exists(hipcc)):
        pytest.fail("toolchain or library missing")
    # the fat binary's gfx950 code object carries the same metadata as notes; extract and read them
    bundler = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        for obj in ("viterbi_kernels.o", "viterbi_lane_kernels.o", "dabplus_kernels.o", "ofdm_kernels.o"):
            src = os.path.join(CSRC, obj)
            if not os.path.exists(src):
                pytest.fail(obj + " not built")
            co = os.path.join(td, obj + ".co")
            # the host object embeds the bundle in section .hip_fatbin
            fat = os.path.join(td, obj + ".fat")
            subprocess.check_call(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", src, fat])
            subprocess.check_call([bundler, "--type=o", "--unbundle", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co])
            notes = subprocess.check_output([objdump, "--notes", co], text=True)
            md = kernel_metadata(notes)
            assert md, obj
            for k, v in md.items():
                assert v["vgpr_spill_count"] == 0 and v["private_segment_fixed_size"] == 0, (obj, k, v)


def test_lane_forward_pass_instruction_count_matches_the_bench_lines_constants(tmp_path):
    """`decoder.roofline` prices the lane decoder's forward pass on two static counts (bench_legs.LANE_VALU_PER_STEP = 202
    VALU instructions per trellis step of one wave, LANE_ACS_VALU_PER_STEP = 96 of them the add-compare-select proper; DESIGN.md
    4.2b).  They are read here from the compiler's listing of lane_forward_grouped_kernel: its 6-step loop body is the
    kernel's largest basic block; 32 `v_pk_max_i16` per step (one per register of packed metric pairs), 64 packed adds /
    subtracts feeding them, and everything else the table in DESIGN.md itemises."""
    import collections
    import bench_legs
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    mk = open(os.path.join(CSRC, "Makefile")).read()
    flags = re.search(r"^CXXFLAGS\s*\?=\s*(.*)$", mk, re.M).group(1).replace("$(ARCH)", "gfx950").replace("$(EXTRA)", "").split()
    out = tmp_path / "viterbi_lane_kernels.s"
    subprocess.check_call([hipcc] + flags + ["-S", "--cuda-device-only", os.path.join(CSRC, "viterbi_lane_kernels.hip"), "-o", str(out)],
                          stderr=subprocess.DEVNULL)
    asm = out.read_text()
    m = re.search(r"^(_ZN4dabk\S*lane_forward_grouped_kernel\S*):[^\n]*\n(.*?)\n\s*s_endpgm", asm, re.S | re.M)
    assert m, "lane_forward_grouped_kernel not found in the listing"
    blocks, cur = [], []
    for line in m.group(2).split("\n"):
        if re.match(r"^\.LBB\d+_\d+:", line):
            blocks.append(cur)
            cur = []
        else:
            cur.append(line)
    blocks.append(cur)
    valu = lambda b: [l.split()[0] for l in b if re.match(r"^\s+v_", l)]
    body = max(blocks, key=lambda b: len(valu(b)))
    ops = collections.Counter(valu(body))
    steps = 6                                                            # one pass through the six phases of the rotating layout
    assert ops["v_pk_max_i16"] == 32 * steps                             # the select of every ACS: exact
    assert ops["v_pk_add_u16"] + ops["v_pk_sub_i16"] >= 64 * steps       # the two candidates of every butterfly (+ branch metrics, renormalisation)
    assert bench_legs.LANE_ACS_VALU_PER_STEP == 32 + 64
    per_step = sum(ops.values()) / steps
    # the constant also carries the per-24-step staging around the body (~5 per step): it may exceed the body's own count by
    # that much and must never be below it
    assert per_step <= bench_legs.LANE_VALU_PER_STEP <= per_step + 8, (per_step, bench_legs.LANE_VALU_PER_STEP)
