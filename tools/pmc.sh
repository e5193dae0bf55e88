#!/bin/bash
# usage (GPU box): tools/pmc.sh <tag> "<counters>" <script> [args]   -- one PMC pass, prints per-kernel means
tag=$1; ctrs=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $root/gpurun_out/pmc_$tag -o $tag -- python3 $root/"$1" "${@:2}" > $root/gpurun_out/pmc_${tag}.log 2>&1
python3 - $root/gpurun_out/pmc_$tag/${tag}_counter_collection.csv <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "dabk" not in k: continue
    short = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in d.items():
        print("   %-28s mean %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
