#!/bin/bash
# GPU box: the bench's timed step with the placed pair and with two plain hipMallocs, in that order and again (two processes
# each): does a box that "behaves as one domain" do better on plain allocations?  usage: tools/box_ab.sh
root=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $root/gpurun_out/box_ab; cd $root
out=$root/gpurun_out/box_ab/$(date +%s).txt
for p in domains plain domains plain; do
python3 bench.py --legs none --steps 8 --placement $p 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); bp, ro = d['config']['buffer_placement'], d['roofline']
print('%-8s domains_seen %s pair/one %s  mover %.3f ms (%.4f)  kernel %.3f ms (%.4f)  decoder %.3f  value %d' % ('$p', bp.get('domains_seen'), bp.get('mover_on_pair_over_mover_in_one_domain'),
      ro['mover_same_geometry_ms'], ro['box_mover_frac'], ro['avg_launch_ms'], ro['frac'], d['decoder']['fic_and_msc_ms'], d['value']))"
done | tee $out
