#!/bin/bash
# GPU box: box_spread's row; on a machine that behaves as one HBM domain whatever is allocated, also tools/domain_matrix.py
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out/box_probe
tools/box_spread.sh > gpurun_out/box_probe/last_row.json
python3 - <<PY
import json,subprocess
d=json.load(open("gpurun_out/box_probe/last_row.json")); rc=d.get("recheck") or {}
chk=rc.get("allocators_own_check") or d.get("pair_over_one_domain") or 0
print("check", chk, "mover_frac", d["box_mover_frac"], "value", d["value"], rc.get("kept"))
if d["box_mover_frac"] < 0.64:
    out=subprocess.run(["python3","tools/domain_matrix.py"],capture_output=True,text=True).stdout
    open("gpurun_out/box_probe/matrix_%d.txt" % d["value"],"w").write(out); print(out[-3000:])
PY
