#!/bin/bash
# GPU job (round 3, first session): GPU test suite, default bench, allocation-kind / CU-mask experiments
tag=${1:-r03_a}
root=${GRAFT_REPO_ROOT:-$(pwd)}
o=$root/gpurun_out/$tag; mkdir -p $o
cd $root
timeout 1500 python3 -m pytest tests -m gpu -x -q > $o/pytest.txt 2>&1; echo "pytest rc=$?" >> $o/pytest.txt
tail -n 5 $o/pytest.txt
timeout 600 python3 bench.py > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"
timeout 600 ./tools/ubench/alloc_kinds 96 > $o/alloc_kinds.txt 2>&1; echo "alloc_kinds rc=$?"
timeout 900 python3 tools/overlap_cumask.py 2>&1 | grep -v amdgpu.ids > $o/overlap_cumask.txt; echo "overlap rc=$?"
cat $o/alloc_kinds.txt | head -40; cat $o/overlap_cumask.txt
python3 - <<PY
import json
j = json.loads(open("$o/bench.json").read().strip().splitlines()[-1])
print({k: j[k] for k in ("value", "ms_per_step", "fic_bit_exact", "msc_bit_exact")}, j["roofline"]["frac"], j.get("sustained"))
PY
