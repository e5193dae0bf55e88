#!/bin/bash
# GPU box: sample power and clocks (rocm-smi, read-only) while the front-end kernel / the data mover run in a loop
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -v "^$" | head -30
echo "== front end in a loop"
python3 tools/ab_inproc.py ofdm 16384 300 10 default > /tmp/pw_ofdm.txt 2>&1 &
pid=$!
sleep 12
for i in 1 2 3 4; do rocm-smi --showpower --showclocks 2>&1 | grep -v WARNING | grep -i "Power (W)\|sclk\|mclk\|fclk" | sed "s/GPU\[0\]\t*: //; s/ clock level//; s/Current Socket Graphics Package //" | tr '\n' ' '; echo; sleep 1; done
wait $pid; grep -v "amdgpu\|^#" /tmp/pw_ofdm.txt | cut -c1-100
echo "== decoder in a loop"
python3 tools/ab_inproc.py decode 16384 300 40 default > /tmp/pw_dec.txt 2>&1 &
pid=$!
sleep 10
for i in 1 2 3; do rocm-smi --showpower --showclocks 2>&1 | grep -v WARNING | grep -i "Power (W)\|sclk\|mclk\|fclk" | sed "s/GPU\[0\]\t*: //; s/ clock level//; s/Current Socket Graphics Package //" | tr '\n' ' '; echo; sleep 1; done
wait $pid; grep -v "amdgpu\|^#" /tmp/pw_dec.txt | cut -c1-100
