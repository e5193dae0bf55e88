#!/usr/bin/env python3
"""Turns the raw files tools/upstream_dump/dab_upstream_dump.h writes into one tests/external/*.npz
(format: INTEGRATION.md section 6; checker: tests/external_vectors.py).

    python tools/upstream_dump/to_npz.py <prefix> <out.npz>

Row t of `msc_<id>` is the logical frame completed by CIF t of the dump (the dumper files every row under its CIF); the
rows before a sub-channel's first one are zero, and `msc_valid_from` = the latest such start over the sub-channels (at
least 15: the 16-CIF de-interleaver)."""
import glob
import os
import sys

import numpy as np


def convert(prefix, out):
    soft = np.fromfile(prefix + ".soft.bin", np.int8)
    n = soft.size // 230400
    if n == 0:
        raise SystemExit("no complete frame in %s.soft.bin" % prefix)
    arrays = {"soft": soft[:n * 230400].reshape(n, 230400)}
    fib = np.fromfile(prefix + ".fib.bin", np.uint8)
    crc = np.fromfile(prefix + ".crc.bin", np.uint8)
    nf = min(n, fib.size // 384, crc.size // 12)
    if nf:
        if nf < n:              # the radio thread lags the OFDM thread by up to two frames: keep what both have
            n = nf
            arrays["soft"] = arrays["soft"][:n]
        arrays["fib"] = fib[:n * 384].reshape(n, 12, 32)
        arrays["crc_ok"] = crc[:n * 12].reshape(n, 12)
    valid_from = 15
    for path in sorted(glob.glob(prefix + ".sub_*.txt")):
        ident = os.path.basename(path)[len(os.path.basename(prefix)) + len(".sub_"):-len(".txt")]
        f = [int(v) for v in open(path).read().split()]
        desc, n_bytes = f[:6], f[6]
        cif = np.fromfile("%s.msc_%s.cif.bin" % (prefix, ident), np.int32)
        rows = np.fromfile("%s.msc_%s.bin" % (prefix, ident), np.uint8)
        rows = rows[:min(rows.size // n_bytes, cif.size) * n_bytes].reshape(-1, n_bytes)
        cif = cif[:rows.shape[0]]
        keep = cif < 4 * n
        if not keep.any():
            continue
        full = np.zeros((4 * n, n_bytes), np.uint8)
        full[cif[keep]] = rows[keep]
        first = int(cif[keep].min())
        if not (np.diff(cif[keep]) == 1).all():
            raise SystemExit("sub-channel %s: its rows are not one per CIF (%s ...): a frame was skipped" % (ident, cif[:8]))
        if int(cif[keep].max()) != 4 * n - 1:
            raise SystemExit("sub-channel %s stops at CIF %d of %d: dump whole frames" % (ident, int(cif[keep].max()), 4 * n))
        arrays["subchannel_" + ident] = np.array(desc, np.int32)
        arrays["msc_" + ident] = full
        valid_from = max(valid_from, first)
    arrays["msc_valid_from"] = np.int32(valid_from)
    if os.path.exists(prefix + ".iq.bin"):
        fo = np.fromfile(prefix + ".fo.bin", np.float32)
        iq = np.fromfile(prefix + ".iq.bin", np.complex64)
        if fo.size:
            per = iq.size // fo.size
            k = min(fo.size, n)
            arrays["iq"] = iq[:k * per].reshape(k, per)
            arrays["freq_offset"] = fo[:k]
    np.savez_compressed(out, **arrays)
    return arrays


if __name__ == "__main__":
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    a = convert(sys.argv[1], sys.argv[2])
    print("wrote %s: %s" % (sys.argv[2], ", ".join("%s%s" % (k, list(np.shape(v))) for k, v in a.items())))
