#!/bin/bash
# GPU box, from the repo root: one profile session of a round -- everything profiles/README.md cites for it.
#   tools/profile_session.sh <tag>        (e.g. r04)      results under gpurun_out/<tag>/
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-$(pwd)}
o=$root/gpurun_out/$tag; mkdir -p $o
cd $root
python3 bench.py > $o/bench.json 2> $o/bench.err                                  # the default line, every leg
tail -c 1500 $o/bench.json
# per-kernel durations of the timed steps alone (no leg after the timed region launches the same kernels)
tools/prof.sh $tag --legs none > $o/prof.log 2>&1
cp gpurun_out/${tag}_kernel_stats.csv $o/kernel_stats.csv; grep "^{" gpurun_out/prof_${tag}_bench.log | tail -n 1 > $o/kernel_stats_bench_line.json
cat $o/kernel_stats.csv
# HBM bytes of the fused front end from the TCC counters (two --pmc passes, calibrated on a 1 GiB copy)
tools/pmc_traffic.sh 16384 cp > $o/pmc_traffic.log 2>&1; cp gpurun_out/pmc_traffic.json $o/pmc_traffic.json; tail -n 8 $o/pmc_traffic.json
# SQ / LDS counters of every kernel of a bench run
tools/pmc_ofdm.sh $tag > $o/pmc_ofdm.log 2>&1; cp gpurun_out/pmc_ofdm_$tag/summary.md $o/pmc_ofdm_summary.md
# the front end's structural variants, the decoder alone, the mover ceiling, the allocator beside a 100 GiB holder
python3 tools/vit_time.py 16384 > $o/vit_time.txt 2>&1; cat $o/vit_time.txt
./tools/ubench/gap_read > $o/gap_read.txt 2>&1; cat $o/gap_read.txt
timeout 900 python3 tools/alloc_stress.py 300 100 > $o/alloc_stress.txt 2>&1; tail -n 2 $o/alloc_stress.txt
python3 tools/frame_latency.py > $o/frame_latency.txt 2>&1; tail -n 3 $o/frame_latency.txt
python3 tools/loop_gate_table.py 4 > $o/loop_gate.txt 2>&1; tail -n 8 $o/loop_gate.txt
python3 tools/single_time.py 1 4 64 256 1024 > $o/single_time.txt 2>&1; cat $o/single_time.txt     # one ensemble: each call of the step, per batch size
python3 tools/decoder_fuzz.py 400 1024 > $o/decoder_fuzz.txt 2>&1; tail -n 1 $o/decoder_fuzz.txt       # the two decoders against each other (and the oracle)
