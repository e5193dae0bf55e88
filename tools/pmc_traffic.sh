#!/bin/bash
# GPU box: HBM traffic of the fused OFDM kernel from the TCC counters, in two PMC passes (FETCH_SIZE needs 3 of the
# 4 TCC slots, WRITE_SIZE 2), calibrated on a 1 GiB device copy in the same process as MI355X_MICROARCH.md prescribes.
n=${1:-1024}; mode=${2:-cp}
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $root/gpurun_out/pmc_$c -o t -- python3 $root/tools/pmc_traffic.py $n $mode > $root/gpurun_out/pmc_$c.log 2>&1
done
python3 - $root $n $mode <<'PY'
import csv, sys, json, collections
root, n, mode = sys.argv[1], int(sys.argv[2]), sys.argv[3]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{root}/gpurun_out/pmc_{c}/t_counter_collection.csv")):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"]
        key = "ofdm" if "ofdm_wave_kernel" in k else ("copy" if ("copy" in k.lower() or "clone" in k.lower()) and float(r["Counter_Value"]) > 1e5 else None)
        if key: acc[key].append(float(r["Counter_Value"]))
    res[c] = {k: sum(v[-3:]) / len(v[-3:]) for k, v in acc.items()}
GiB = 1024 ** 3
# counters are in KiB; calibrate on the copy (1 GiB read, 1 GiB written)
rd_scale = GiB / (res["FETCH_SIZE"]["copy"] * 1024)
wr_scale = GiB / (res["WRITE_SIZE"]["copy"] * 1024)
rd = res["FETCH_SIZE"]["ofdm"] * 1024 * rd_scale
wr = res["WRITE_SIZE"]["ofdm"] * 1024 * wr_scale
alg = ((76 * 2048 + 504) * 8 + 230400) if mode == "dd" else 1782016
out = {"frames_per_launch": n, "mode": "decision-directed, no cyclic prefix read" if mode == "dd" else "with cyclic-prefix correlations", "raw_kib": res, "read_scale_from_1GiB_copy": rd_scale, "write_scale_from_1GiB_copy": wr_scale,
       "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
       "algorithmic_bytes_per_launch": alg * n, "ratio_to_algorithmic": (rd + wr) / (alg * n)}
json.dump(out, open(f"{root}/gpurun_out/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
