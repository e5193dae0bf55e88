#!/bin/bash
# GPU box: power and clocks while the exact-geometry data mover runs (tools/ubench/frame_layout, noise input)
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
( for k in 1 2 3; do ./tools/ubench/frame_layout > /tmp/fl_$k.txt 2>&1; done ) &
pid=$!
sleep 9
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks 2>&1 | grep -v WARNING | grep -i "Power (W)\|sclk" | sed "s/GPU\[0\]\t*: //; s/ clock level//; s/Current Socket Graphics Package //" | tr '\n' ' '; echo; sleep 1.5; done
wait $pid; head -3 /tmp/fl_2.txt
