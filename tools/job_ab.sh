#!/bin/bash
# GPU job: in-process A/B of the in-tree library against build/ab/libdabgpu_base.so (modes in $2), tag = $1
tag=${1:-r03_ab}; modes=${2:-"ofdm fft"}
root=${GRAFT_REPO_ROOT:-$(pwd)}
o=$root/gpurun_out/$tag; mkdir -p $o
cd $root
for m in $modes; do python3 tools/ab_inproc.py $m 16384 5 3 build/ab/libdabgpu_base.so default 2>&1 | grep -v "amdgpu.ids"; done > $o/ab.txt 2>&1
cat $o/ab.txt
