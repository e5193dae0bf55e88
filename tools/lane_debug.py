#!/usr/bin/env python3
"""Compare the lane-per-codeword Viterbi with the oracle on noise (GPU box; test helper, imports oracle)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import numpy as np, dabgpu
from oracle import oracle as O
ctx = dabgpu.Context(0, 16, flags=dabgpu.FLAG_VITERBI_LANE)
rng = np.random.default_rng(4)
noise = rng.integers(-127, 128, size=(9, 9216), dtype=np.int8)
noise[7] = 0
noise[8] = rng.choice(np.array([-127, 127], np.int8), 9216)
fib, ok = ctx.fic_decode(noise)
for f in range(9):
    ofib, ook = O.fic_decode(noise[f])
    ofib = np.asarray(ofib).reshape(12, 32); g = np.asarray(fib[f]).reshape(12, 32)
    for grp in range(4):
        a = np.unpackbits(g[3 * grp:3 * grp + 3].reshape(-1)); b = np.unpackbits(ofib[3 * grp:3 * grp + 3].reshape(-1))
        bad = np.nonzero(a != b)[0]
        if len(bad): print("frame", f, "group", grp, "nbad", len(bad), "first", bad[:8], "last", bad[-4:])
print("done")
