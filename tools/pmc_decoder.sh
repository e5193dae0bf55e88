#!/bin/bash
# GPU box: HBM traffic of the channel decoder's kernels from the TCC counters (two --pmc passes, calibrated on a 1 GiB
# device copy in the same process, as MI355X_MICROARCH.md prescribes).  usage: tools/pmc_decoder.sh [n_frames]
n=${1:-16384}
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $root/gpurun_out/pmcdec_$c -o t -- python3 $root/tools/pmc_decoder.py $n > $root/gpurun_out/pmcdec_$c.log 2>&1
done
python3 - $root $n <<'PY'
import csv, sys, json, collections
root, n = sys.argv[1], int(sys.argv[2])
res, dur = {}, collections.defaultdict(list)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{root}/gpurun_out/pmcdec_{c}/t_counter_collection.csv")):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"]; v = float(r["Counter_Value"])
        if "dabk" in k:
            acc[k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]].append(v)
        elif ("copy" in k.lower() or "clone" in k.lower()) and v > 1e5:
            acc["copy"].append(v)
    res[c] = {k: sum(v[-3:]) / len(v[-3:]) for k, v in acc.items()}
    for r in csv.DictReader(open(f"{root}/gpurun_out/pmcdec_{c}/t_kernel_trace.csv")):
        k = r["Kernel_Name"]
        if "dabk" in k:
            dur[k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
GiB = 1024 ** 3
rs, ws = GiB / (res["FETCH_SIZE"]["copy"] * 1024), GiB / (res["WRITE_SIZE"]["copy"] * 1024)
out = {"frames_per_call": n, "workload": "dabgpu_decode_frames_dev: FIC + one 64 kbit/s EEP 3-A sub-channel, 64 streams, noise soft bits, history carried",
       "read_scale_from_1GiB_copy": rs, "write_scale_from_1GiB_copy": ws, "kernels": {}}
# algorithmic bytes of the call: A_fic = 9216 + 396, A_msc64 = 37 632 per frame; survivors (the lane decoder's
# own scratch, written once and read once): 8 B x 64 states... = steps x 8 B per codeword, FIC 774 + sub-channel 1542, 4 + 4 codewords
alg = {"survey_A_fic_plus_A_msc64_per_frame": 9612 + 37632, "survivor_bytes_written_per_frame": 8 * 4 * (774 + 1542)}
for k in sorted(set(res["FETCH_SIZE"]) | set(res["WRITE_SIZE"])):
    if k == "copy": continue
    rd = res["FETCH_SIZE"].get(k, 0.0) * 1024 * rs; wr = res["WRITE_SIZE"].get(k, 0.0) * 1024 * ws
    d = dur.get(k, [])
    ms = sum(d[-6:]) / max(1, len(d[-6:]))
    out["kernels"][k] = {"hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "bytes_per_frame": (rd + wr) / n,
                         "avg_ms_under_the_profiler": ms, "GBps": (rd + wr) / (ms * 1e-3) / 1e9 if ms else None}
out["algorithmic"] = alg
out["total_bytes_per_frame"] = sum(v["bytes_per_frame"] for v in out["kernels"].values())
json.dump(out, open(f"{root}/gpurun_out/pmc_decoder.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
