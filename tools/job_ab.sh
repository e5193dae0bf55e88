#!/bin/bash
# GPU job: A/B of library variants in one session.  usage: tools/job_ab.sh <tag> <tool.py> "<tool args>" <rounds> name1 name2 ...
# (variant "default" = the in-tree libdabgpu.so, otherwise build/ab/libdabgpu_<name>.so)
tag=$1; tool=$2; args=$3; rounds=$4; shift 4
root=${GRAFT_REPO_ROOT:-$(pwd)}; o=$root/gpurun_out/$tag; mkdir -p $o; cd $root
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    if [ "$v" = default ]; then unset DABGPU_LIB; else export DABGPU_LIB=$root/build/ab/libdabgpu_$v.so; fi
    echo "== round $r variant $v" >> $o/ab.txt
    python3 $tool $args 2>&1 | grep -v amdgpu.ids >> $o/ab.txt
  done
done
cat $o/ab.txt
