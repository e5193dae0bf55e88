#!/bin/bash
# GPU box: the grouped decoder (FIC + one sub-channel, 16 384 frames) under several builds of the lane kernels, each in
# processes of its own, twice, interleaved: build/ab/libdabgpu_<name>.so (tools/lane_ab.sh name1 name2 ...)
root=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
  for n in "$@"; do
    DABGPU_LIB=$root/build/ab/libdabgpu_$n.so python3 $root/tools/vit_time.py 16384 2>/dev/null | head -2 | tr '\n' ' ' | sed "s|$root/build/ab/||"; echo
  done
done
