#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof.sh <tag> [bench args...]
# runs rocprofv3 --kernel-trace --stats on bench.py and writes the trimmed summary to gpurun_out/<tag>_kernel_stats.csv
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
# --no-traffic: bench.py must never start its own rocprofv3 children from under a profiler (it also refuses by itself)
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -o $tag -- python3 $root/bench.py --no-traffic "$@" > $root/gpurun_out/prof_${tag}_bench.log 2>&1
python3 $root/tools/rocprof_summary.py $root/gpurun_out/prof_$tag/${tag}_kernel_stats.csv $root/gpurun_out/${tag}_kernel_stats.csv "python3 bench.py $*"
cat $root/gpurun_out/${tag}_kernel_stats.csv
