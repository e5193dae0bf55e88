#!/bin/bash
# GPU job: default bench line, rocprofv3 kernel stats, HBM traffic PMC passes, SQ PMC passes  (tag = $1)
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-$(pwd)}
o=$root/gpurun_out/$tag; mkdir -p $o
cd $root
python3 bench.py > $o/bench.json 2> $o/bench.err
tail -c 4000 $o/bench.json
tools/prof.sh $tag --cpu-seconds 0 --no-closed-loop --no-sustained --no-selective --no-cp-leg --no-plain-compare > $o/prof.log 2>&1
cp gpurun_out/${tag}_kernel_stats.csv $o/kernel_stats.csv
tools/pmc_traffic.sh 16384 dd > $o/pmc_traffic.log 2>&1
cp gpurun_out/pmc_traffic.json $o/pmc_traffic.json
tools/pmc_ofdm.sh $tag > $o/pmc_ofdm.log 2>&1
cp gpurun_out/pmc_ofdm_$tag/summary.md $o/pmc_ofdm_summary.md
cat $o/kernel_stats.csv; cat $o/pmc_traffic.json | tail -8
./tools/ubench/read_bw > $o/read_bw.txt 2>&1
./tools/ubench/gap_read > $o/gap_read.txt 2>&1
python3 tools/frame_latency.py > $o/frame_latency.txt 2>&1
python3 tools/pcie_rate.py 1 > $o/pcie_rate.txt 2>&1; python3 tools/pcie_rate.py 64 >> $o/pcie_rate.txt 2>&1
python3 tools/ensemble_time.py > $o/ensemble_time.txt 2>&1
python3 tools/ofdm_bound.py 16384 5 > $o/ofdm_bound.txt 2>&1
tail -n 3 $o/frame_latency.txt $o/pcie_rate.txt $o/ensemble_time.txt; cat $o/ofdm_bound.txt; grep -v "^#" $o/read_bw.txt | tail -n 12; cat $o/gap_read.txt
