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
# the same with page-locked buffers from dabgpu_host_alloc on both sides (C ABI called directly, outputs in place)
import ctypes as C
L = dabgpu.lib()
p_iq = dabgpu.PinnedArray(iq.shape, np.complex64); p_iq.array[:] = iq
p_soft = dabgpu.PinnedArray((n, dabgpu.NB_FRAME_BITS), np.int8)
p_fib = dabgpu.PinnedArray((n, 12, 32), np.uint8); p_ok = dabgpu.PinnedArray((n, 12), np.uint8)
def call():
    assert L.dabgpu_ofdm_demod_frames(ctx._h, p_iq.array.ctypes.data, iq.shape[1], n, fo.ctypes.data, p_soft.array.ctypes.data, None, None) == 0
    assert L.dabgpu_fic_decode(ctx._h, p_soft.array.ctypes.data, dabgpu.NB_FRAME_BITS, n, p_fib.array.ctypes.data, p_ok.array.ctypes.data) == 0
for _ in range(2): call()
t0 = time.perf_counter()
for _ in range(reps): call()
dt = (time.perf_counter() - t0) / reps
assert (p_soft.array == soft).all() and (p_fib.array == fib).all()
print("page-locked buffers : %d frames per call, %.2f ms per call, %.0f frames/s (%.1f GB/s of IQ over PCIe)" % (
    n, dt * 1e3, n / dt, n * 76 * 2552 * 8 / dt / 1e9))
