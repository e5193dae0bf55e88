#!/bin/bash
# usage (GPU box, repo root): tools/prof_any.sh <tag> <script.py> [args]   -- per-kernel times of any tool script
tag=$1; script=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -o $tag -- python3 $root/$script "$@" > $root/gpurun_out/prof_${tag}.log 2>&1
python3 $root/tools/rocprof_summary.py $root/gpurun_out/prof_$tag/${tag}_kernel_stats.csv $root/gpurun_out/${tag}_kernel_stats.csv "python3 $script $*"
cat $root/gpurun_out/${tag}_kernel_stats.csv
