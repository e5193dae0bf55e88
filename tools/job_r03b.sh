#!/bin/bash
# GPU job: a subset of the GPU tests (first argument: pytest selection), then optionally the bench
tag=${1:-r03_b}; sel=${2:-tests}; bench=${3:-no}
root=${GRAFT_REPO_ROOT:-$(pwd)}
o=$root/gpurun_out/$tag; mkdir -p $o
cd $root
timeout 2400 python3 -m pytest $sel -m gpu -x -q > $o/pytest.txt 2>&1; echo "pytest rc=$?" >> $o/pytest.txt
tail -n 60 $o/pytest.txt
if [ "$bench" != "no" ]; then
  timeout 900 python3 bench.py $bench > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"; tail -n 5 $o/bench.err
  python3 - <<PY
import json
j = json.loads(open("$o/bench.json").read().strip().splitlines()[-1])
print({k: j[k] for k in ("value", "ms_per_step", "fic_bit_exact", "msc_bit_exact")}, j["roofline"]["frac"])
print(json.dumps(j.get("closed_loop"), indent=1))
PY
fi
