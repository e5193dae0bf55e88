#!/bin/bash
# GPU job: micro-benchmarks + bound study + PMC passes (round 2, session 1)
root=${GRAFT_REPO_ROOT:-$(pwd)}
o=$root/gpurun_out/r02_s1; mkdir -p $o
cd $root
rocm-smi --showclocks > $o/clocks.txt 2>&1
./tools/ubench/valu_cycles > $o/valu_cycles.txt 2>&1
./tools/ubench/lds_conflict > $o/lds_conflict.txt 2>&1
( cd /tmp; export TMPDIR=/tmp
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_WAVE_CYCLES \
    --kernel-trace --output-format csv -d $o/pmc_lds -o lds -- $root/tools/ubench/lds_conflict > $o/pmc_lds.log 2>&1 )
python3 - $o/pmc_lds/lds_counter_collection.csv > $o/lds_conflict_pmc.txt <<'PY'
import csv, sys, collections
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    acc.setdefault(k, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    # second launch of each kernel = the long one
    v = {c: x[-1] for c, x in d.items()}
    print("%-40s IDX_ACTIVE %.4g BANK_CONFLICT %.4g ratio %.3f INSTS_LDS %.4g -> %.2f array cycles/inst ; ADDR_CONFLICT %.3g UNALIGNED %.3g" % (
        k[:40], v["SQ_LDS_IDX_ACTIVE"], v["SQ_LDS_BANK_CONFLICT"], v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1),
        v["SQ_INSTS_LDS"], v["SQ_LDS_IDX_ACTIVE"] / max(v["SQ_INSTS_LDS"], 1), v["SQ_LDS_ADDR_CONFLICT"], v["SQ_LDS_UNALIGNED_STALL"]))
PY
for r in 1 2; do
  for v in base noclamp twtab both; do
    DABGPU_LIB=$root/build/ab/libdabgpu_$v.so python3 tools/ofdm_bound.py 16384 5 >> $o/bound.txt 2>> $o/bound.err
  done
done
tools/pmc_ofdm.sh r02_base > $o/pmc_ofdm.log 2>&1
cp $root/gpurun_out/pmc_ofdm_r02_base/summary.md $o/pmc_ofdm_base_summary.md
tail -n 40 $o/valu_cycles.txt; cat $o/lds_conflict.txt $o/lds_conflict_pmc.txt $o/bound.txt
