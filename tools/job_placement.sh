#!/bin/bash
# GPU job: evidence for the buffer-placement effect on the HBM-bound front end (tag = $1)
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-$(pwd)}
o=$root/gpurun_out/$tag; mkdir -p $o
cd $root
{
  echo "## tools/ubench/placement (data mover, hipMalloc'ed buffers, one process)"
  timeout 600 ./tools/ubench/placement 4
  echo
  echo "## tools/placement.py (the fused OFDM kernel, torch allocations, one process)"
  python3 tools/placement.py 16384 3 2>&1 | grep -v amdgpu.ids
  echo
  echo "## tools/ubench/frame_layout (data mover with the kernel's exact geometry; first 4 lines of each buffer pair), run twice"
  timeout 600 ./tools/ubench/frame_layout | grep -A3 "buffer pair"
  timeout 600 ./tools/ubench/frame_layout | grep -A3 "buffer pair"
  echo
  echo "## tools/ubench/cache_policy (sc0 / nt / sc1 bits on the mover's loads and stores)"
  timeout 600 ./tools/ubench/cache_policy
} > $o/placement.txt 2>&1
# (build/ab/libdabgpu_base.so: the library as of commit 122bffd, i.e. before the streaming accesses -- check that
# commit out into a scratch tree, `make -C sdrplusplus-dab-radio-plugin_amd/csrc`, copy libdabgpu.so there)
for m in ofdm fft select acquire decode multiplex; do python3 tools/ab_inproc.py $m 16384 5 3 build/ab/libdabgpu_base.so default 2>&1 | grep -v "amdgpu.ids"; done > $o/ab_nontemporal.txt 2>&1
cat $o/placement.txt | tail -n 50; cat $o/ab_nontemporal.txt | grep -v "^#"
