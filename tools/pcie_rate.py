#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry points (never the bench `value`): frames/s when IQ starts in host
memory and soft bits / FIBs come back to host memory, one call at a time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import numpy as np, dabgpu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(0)
iq = (rng.standard_normal((n, 76 * 2552)) + 1j * rng.standard_normal((n, 76 * 2552))).astype(np.complex64)
fo = np.zeros(n, np.float32)
ctx = dabgpu.Context(0, n)
for _ in range(2):
    soft, _, _ = ctx.ofdm_demod_frames(iq, fo); ctx.fic_decode(soft)
t0 = time.perf_counter(); reps = 5
for _ in range(reps):
    soft, _, _ = ctx.ofdm_demod_frames(iq, fo)
    fib, ok = ctx.fic_decode(soft)
dt = (time.perf_counter() - t0) / reps
print("host-buffer path: %d frames per call, %.2f ms per call, %.0f frames/s (%.1f GB/s of IQ over PCIe)" % (
    n, dt * 1e3, n / dt, n * 76 * 2552 * 8 / dt / 1e9))
