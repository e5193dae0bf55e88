#!/bin/bash
# Build A/B variants of libdabgpu.so that differ only in ofdm_kernels.hip compile flags:
#   tools/build_variants.sh name1 "flags1" name2 "flags2" ...   -> build/ab/libdabgpu_<name>.so
# (build/ is git-ignored; the .so files travel to the GPU box with the snapshot)
set -e
root=$(cd $(dirname $0)/.. && pwd)
src=$root/sdrplusplus-dab-radio-plugin_amd/csrc
out=$root/build/ab
mkdir -p $out
make -s -C $src -j6
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -Wall -Wno-unused-function"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc $FLAGS $flags -c $src/ofdm_kernels.hip -o $out/ofdm_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libdabgpu_$name.so $out/ofdm_$name.o $src/dabgpu_api.o \
       $src/viterbi_kernels.o $src/viterbi_lane_kernels.o $src/sync_kernels.o $src/dabplus_kernels.o && echo built $name ) &
done
wait
