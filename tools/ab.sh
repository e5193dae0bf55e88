#!/bin/bash
# usage (GPU box): tools/ab.sh <timing script> "<flags of variant B>" [rounds]
# Builds variant B (EXTRA flags) next to the default library and alternates A/B runs (the pool boxes' clocks drift
# by several percent between runs): prints every run and lets the reader take the minimum per variant.
cd $GRAFT_REPO_ROOT
script=$1; flags=$2; rounds=${3:-4}
mkdir -p /tmp/objab
make -s -C sdrplusplus-dab-radio-plugin_amd/csrc OBJDIR=/tmp/objab OUT=/tmp/libab.so EXTRA="$flags" 2>&1 | grep -E "error"
for r in $(seq $rounds); do
  echo -n "A "; python $script 2>/dev/null | tail -1
  echo -n "B "; DABGPU_LIB=/tmp/libab.so python $script 2>/dev/null | tail -1
done
