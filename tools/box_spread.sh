#!/bin/bash
# GPU box: one line for profiles/r05_box_spread.txt -- what THIS fresh box gives the bench's timed step:
# how many HBM domains the placement probe saw, the mover's ceiling, the kernel, the step.
# usage (one gpurun call = one fresh box): tools/box_spread.sh
root=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $root/gpurun_out/box_spread; cd $root
python3 bench.py --legs none --steps 10 2>/dev/null | python3 -c "
import json, sys, time, socket
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
bp, ro = d['config']['buffer_placement'], d['roofline']
row = {'host': socket.gethostname(), 'domains_seen': bp.get('domains_seen'), 'method': bp['method'][:12], 'fallback': bp.get('fallback_reason'),
       'same_domain_per_mille': bp.get('soft_bits_written_beside_same_domain_reads_per_mille'),
       'pair_over_one_domain': bp.get('mover_on_pair_over_mover_in_one_domain'),
       'mover_ms': round(ro['mover_same_geometry_ms'], 3), 'box_mover_frac': round(ro['box_mover_frac'], 4), 'kernel_ms': round(ro['avg_launch_ms'], 3),
       'frac': round(ro['frac'], 4), 'kernel_over_mover': round(ro['kernel_over_mover'], 3), 'decoder_ms': round(d['decoder']['fic_and_msc_ms'], 3),
       'recheck': bp.get('placed_vs_plain') or bp.get('one_domain_box_recheck'), 'value': round(d['value']), 'step_ms': {k: round(v, 3) for k, v in d['step_ms'].items() if k != 'what'}}
print(json.dumps(row))
" | tee $root/gpurun_out/box_spread/$(date +%s).json
