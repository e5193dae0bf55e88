#!/usr/bin/env python3
"""Turn the per-group PMC CSVs of tools/pmc_ofdm.sh into a one-page summary (markdown on stdout).

Units (MI355X_MICROARCH.md, "Per-instruction cycle constants"): SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count
quad-cycles (4 shader cycles); SQ_BUSY_CYCLES counts cycles per shader engine (32 SEs summed), GRBM_GUI_ACTIVE cycles
per XCD (8 summed); SQ_INSTS_* count wave-instructions; SQ_LDS_* count LDS-array cycles summed over CUs."""
import collections
import csv
import glob
import os
import sys

N_SIMD, N_CU, N_XCD = 1024, 256, 8
SYMS_PER_FRAME_FUSED = 76 + 2          # 3 runs per frame: 2 reference symbols transformed twice


def load(root):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(root, "g*", "g*_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "dabk" not in k:
                continue
            short = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in sorted(glob.glob(os.path.join(root, "g*", "g*_kernel_trace.csv"))):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "dabk" not in k:
                continue
            short = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            dur[short].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
    return acc, dur


def main():
    root = sys.argv[1]
    n_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    acc, dur = load(root)
    for k in sorted(acc):
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        t = sum(dur[k]) / max(len(dur[k]), 1)
        print("## `%s`\n" % k)
        print("mean of %d profiled launches, %.3f ms per launch (profiled runs clock lower than unprofiled ones)\n" % (len(dur[k]), t * 1e3))
        print("| counter | mean per launch |\n|---|---|")
        for n in sorted(c):
            print("| %s | %.4g |" % (n, c[n]))
        print()
        g = c.get
        if g("GRBM_GUI_ACTIVE") and t:
            cyc = g("GRBM_GUI_ACTIVE") / N_XCD
            print("* kernel duration %.3g shader cycles per XCD -> effective clock %.0f MHz" % (cyc, cyc / t / 1e6))
            if g("SQ_INSTS_VALU"):
                per_simd = g("SQ_INSTS_VALU") / N_SIMD
                print("* VALU: %.4g wave-instructions = %.4g per SIMD = %.2f shader cycles of kernel time per VALU instruction per SIMD"
                      % (g("SQ_INSTS_VALU"), per_simd, cyc / per_simd))
                if "ofdm_wave_kernel" in k:
                    n_sym = n_frames * (SYMS_PER_FRAME_FUSED if "<false" in k else 76)
                    print("* = %.0f VALU, %.0f LDS, %.0f scalar instructions per symbol-wave (%d symbol transforms)" % (
                        g("SQ_INSTS_VALU") / n_sym, g("SQ_INSTS_LDS", 0) / n_sym, g("SQ_INSTS_SALU", 0) / n_sym, n_sym))
            if g("SQ_ACTIVE_INST_VALU"):
                print("* SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = %.3f quad-cycles per instruction; x4 / (SIMDs x kernel cycles) = %.1f %% "
                      "(an UPPER bound on VALU occupancy: the counter ticks one quad-cycle per instruction, also for 2-cycle ones -- "
                      "see the valu_cycles micro-benchmark)" % (g("SQ_ACTIVE_INST_VALU") / g("SQ_INSTS_VALU"),
                                                                 400.0 * g("SQ_ACTIVE_INST_VALU") / (N_SIMD * cyc)))
        if g("SQ_WAVE_CYCLES"):
            w = g("SQ_WAVE_CYCLES")
            print("* wave-cycle split (quad-cycles): " + ", ".join(
                "%s %.1f %%" % (n, 100 * g(n) / w) for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS") if g(n)))
        if g("SQ_LDS_IDX_ACTIVE"):
            print("* LDS: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = %.3f; LDS array busy %.1f %% of CU-cycles"
                  % (g("SQ_LDS_BANK_CONFLICT", 0) / g("SQ_LDS_IDX_ACTIVE"),
                     100.0 * g("SQ_LDS_IDX_ACTIVE") / (N_CU * g("GRBM_GUI_ACTIVE") / N_XCD) if g("GRBM_GUI_ACTIVE") else float("nan")))
            if "ofdm_wave_kernel" in k:
                n_sym = n_frames * (SYMS_PER_FRAME_FUSED if "<false" in k else 76)
                print("* = %.0f LDS-array cycles per symbol-wave, %.0f of them bank-conflict cycles" % (
                    g("SQ_LDS_IDX_ACTIVE") / n_sym, g("SQ_LDS_BANK_CONFLICT", 0) / n_sym))
        print()


if __name__ == "__main__":
    main()
