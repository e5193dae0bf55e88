#!/usr/bin/env python3
"""GPU box: one DAB+ super-frame per call through dabgpu_dabplus_superframes on page-locked buffers (what the host mirror's
channels do 2.4 times per frame): wall clock per call; under `rocprofv3 --kernel-trace --stats` the kernel's own time.
A library built with EXTRA=-DDABPLUS_PHASE_TIMING leaves the kernel's phase stamps in status.reserved (shader-clock ticks).
usage: tools/dabplus_time.py [bitrate ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np, dabgpu
from dabgpu import synth
rates = [int(a) for a in sys.argv[1:]] or [64, 48, 32, 128]
if os.environ.get("DABPLUS_TIME_LIB"):                     # another build of the library (the phase-stamped one)
    dabgpu.LIB_PATH = os.environ["DABPLUS_TIME_LIB"]
ctx = dabgpu.Context(device=0, max_frames=1)
lib = dabgpu.lib()
ST = np.dtype([("firecode_ok", np.int32), ("rs_corrected", np.int32), ("rs_uncorrectable", np.int32), ("num_aus", np.int32),
               ("au_crc_mask", np.int32), ("au_start", np.int32, (8,)), ("reserved", np.int32, (3,))])
for br in rates:
    s = br // 8
    rng = np.random.default_rng(br)
    sf, starts, aus = synth.build_superframe(rng, br, 1, 1)
    p_in = dabgpu.PinnedArray((120 * s,), np.uint8); p_in.array[:] = sf
    p_out = dabgpu.PinnedArray((110 * s,), np.uint8)
    p_st = dabgpu.PinnedArray((1,), ST)
    def call():
        rc = lib.dabgpu_dabplus_superframes(ctx._h, p_in.array.ctypes.data, 120 * s, 1, br, p_out.array.ctypes.data, p_st.array.ctypes.data)
        assert rc == 0
    for _ in range(50): call()
    n = 1000
    t0 = time.perf_counter()
    for _ in range(n): call()
    dt = (time.perf_counter() - t0) / n
    st = p_st.array[0]
    ok = bool(st["firecode_ok"]) and st["au_crc_mask"] == (1 << st["num_aus"]) - 1 and (p_out.array == sf[:110 * s]).all()
    print("%3d kbit/s: %.1f us per call (wall); super-frame clean: %s; %d access units; reserved %s" % (br, dt * 1e6, ok, st["num_aus"], list(st["reserved"])))
