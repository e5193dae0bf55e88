#!/bin/bash
# GPU box: PMC passes over the bench's fused front end (one rocprofv3 --pmc run per counter group, nothing but
# --kernel-trace beside it), then tools/pmc_ofdm_summary.py turns the means into the one-page summary kept under
# profiles/.  usage: tools/pmc_ofdm.sh <tag> [bench args...]
tag=${1:-r02}; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_ofdm_$tag
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
groups=(
 "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"
 "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"
)
i=0
for g in "${groups[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out/g$i -o g$i -- python3 $root/bench.py --legs none --steps 4 --warmup 1 "$@" > $out/g$i.log 2>&1
  echo "group $i rc $?"
done
python3 $root/tools/pmc_ofdm_summary.py $out > $out/summary.md
cat $out/summary.md
rm -rf $out/g[1-9]        # the raw per-dispatch rows (60 MB): gpurun_out/ may carry 64 MiB home
