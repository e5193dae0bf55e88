"""The C-ABI library loads on a GPU-less host, exports every symbol include/dabgpu.h
declares, and refuses to compute without a gfx950 device (no CPU fallback)."""
import os
import re

import pytest

import dabgpu
from conftest import ROOT


def test_header_symbols_are_all_exported(built):
    hdr = open(os.path.join(ROOT, "include", "dabgpu.h")).read()
    declared = set(re.findall(r"\b(dabgpu_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(dabgpu.EXPORTS)
    L = dabgpu.lib()
    for sym in declared:
        assert hasattr(L, sym), sym
    assert L.dabgpu_abi_version() == dabgpu.ABI_VERSION == 6
    # the binding declares every entry point's argument types (an undeclared pointer argument would be passed as a C int)
    assert [n for n in dabgpu.EXPORTS if getattr(L, n).argtypes is None] == []


def test_header_is_plain_c(tmp_path):
    """include/dabgpu.h is what a C (cgo / JNI / ctypes-generator) binding compiles against: it must be valid C99 and
    C++11 on its own, warning-free."""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('#include "dabgpu.h"\n'
                   'int main(void) { dabgpu_cfg c = {0, 1, 1, 0, 0, {0, 0, 0}}; dabgpu_subchannel s = {0, 48, 0, 0, 3, 64};\n'
                   '  dabgpu_bit_range r[9]; (void)c; return dabgpu_soft_selection(&s, 1, 1, r, 9) == 5 ? 0 : 1; }\n')
    inc = "-I" + os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", inc, str(src)])
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c++", inc, str(src)])


def test_strerror_and_argument_errors(built):
    assert dabgpu.strerror(0) == "ok"
    assert "gfx950" in dabgpu.strerror(-4)
    L = dabgpu.lib()
    assert L.dabgpu_get_ofdm_params(1, None) == -1
    assert L.dabgpu_create(None, None) == -1
    sc = dabgpu.subchannel(0, 64, level=3)
    assert L.dabgpu_subchannel_bytes(sc) == 192
    bad = dabgpu.Subchannel(0, 47, 0, 0, 3, 64)           # length does not match the profile
    assert L.dabgpu_subchannel_bytes(bad) == -5
    assert L.dabgpu_subchannel_bytes(dabgpu.Subchannel(850, 48, 0, 0, 3, 64)) == -1   # beyond CU 863


def test_closed_loop_defaults_are_the_reference_estimator(built):
    """every closed-loop entry point defaults to the loop the reference has -- the cyclic-prefix
    correlations (fine_freq_update_beta, /root/reference/src/render_radio_block.cpp:216); this library's decision-directed
    estimator is the opt-in, in the ABI (dabgpu_track_cfg.decision_directed, dabgpu_set_stream_loop) and in the host
    mirror (OFDM_Demod_Config.sync.is_decision_directed_fine_freq)."""
    assert dabgpu.track_cfg().decision_directed == 0
    assert dabgpu.track_cfg().dd_gate == 2.5
    import inspect
    assert inspect.signature(dabgpu.Context.set_stream_loop).parameters["decision_directed"].default is False
    hdr = open(os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd", "host", "ofdm", "ofdm_demodulator.h")).read()
    assert "bool is_decision_directed_fine_freq = false;" in hdr
    src = open(os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd", "host", "ofdm", "ofdm_demodulator.cpp")).read()
    assert "cfg.decision_directed = knob(m_cfg.sync.is_decision_directed_fine_freq) ? 1 : 0;" in src


def test_no_cpu_fallback(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(dabgpu.DabGpuError) as e:
        dabgpu.Context(device=0)
    assert e.value.status == -4


def test_product_path_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "dab_oracle" not in txt and "liboracle" not in txt, os.path.join(dirpath, f)
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, re.M), os.path.join(dirpath, f)


def test_binding_structures_match_the_header(tmp_path):
    """Every ctypes structure of the binding has the size the C compiler gives the header's struct, and its fields sit at
    the header's offsets (a field added to one side only would shift everything behind it silently)."""
    import ctypes as C
    import subprocess
    pairs = [("dabgpu_ofdm_params", dabgpu.OfdmParams), ("dabgpu_dab_params", dabgpu.DabParams), ("dabgpu_cfg", dabgpu.Cfg),
             ("dabgpu_stats", dabgpu.Stats), ("dabgpu_sync_result", dabgpu.SyncResult), ("dabgpu_acquire_cfg", dabgpu.AcquireCfg),
             ("dabgpu_track_cfg", dabgpu.TrackCfg), ("dabgpu_placement_report", dabgpu.PlacementReport),
             ("dabgpu_frame_result", dabgpu.FrameResult), ("dabgpu_subchannel", dabgpu.Subchannel)]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "dabgpu.h"', 'int main(void) {']
    for cname, cls in pairs:
        lines.append('  printf("%s %%zu", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('  printf(" %%zu", offsetof(%s, %s));' % (cname, fname))
        lines.append('  printf("\\n");')
    # the two device-resident records the tests read through numpy dtypes
    records = [("dabgpu_stream_state", dabgpu.STREAM_STATE_DTYPE), ("dabgpu_acquired_frame", dabgpu.ACQUIRED_FRAME_DTYPE)]
    for cname, dt in records:
        lines.append('  printf("%s %%zu", sizeof(%s));' % (cname, cname))
        for fname in dt.names:
            lines.append('  printf(" %%zu", offsetof(%s, %s));' % (cname, fname))
        lines.append('  printf("\\n");')
    lines += ['  return 0; }']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines) + "\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True).split("\n")
    for (cname, cls), line in zip(pairs, out):
        got = line.split()
        assert got[0] == cname
        want = [C.sizeof(cls)] + [getattr(cls, f).offset for f, _ in cls._fields_]
        assert [int(x) for x in got[1:]] == want, (cname, got[1:], want)
    for (cname, dt), line in zip(records, out[len(pairs):]):
        got = line.split()
        assert got[0] == cname
        assert [int(x) for x in got[1:]] == [dt.itemsize] + [dt.fields[f][1] for f in dt.names], (cname, got[1:])
    assert dabgpu.STREAM_STATE_DTYPE.itemsize == 64 and dabgpu.ACQUIRED_FRAME_DTYPE.itemsize == 32
