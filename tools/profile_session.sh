#!/bin/bash
# GPU box, from the repo root: one profile session of a round -- everything profiles/README.md cites for it.
#   tools/profile_session.sh <tag>        (e.g. r06)      results under gpurun_out/<tag>/
tag=${1:-r06}
root=${GRAFT_REPO_ROOT:-$(pwd)}
o=$root/gpurun_out/$tag; mkdir -p $o
cd $root
python3 bench.py > $o/bench.json 2> $o/bench.err                                  # the default line, every leg
tail -c 1500 $o/bench.json
# per-kernel durations of the timed steps alone (no leg after the timed region launches the same kernels)
tools/prof.sh $tag --legs none > $o/prof.log 2>&1
cp gpurun_out/${tag}_kernel_stats.csv $o/kernel_stats.csv; grep "^{" gpurun_out/prof_${tag}_bench.log | tail -n 1 > $o/kernel_stats_bench_line.json
cat $o/kernel_stats.csv
# HBM bytes of the fused front end from the TCC counters (two --pmc passes, calibrated on a 1 GiB copy): the timed step's
# data flow (cyclic-prefix correlations) and the opt-in decision-directed one
tools/pmc_traffic.sh 16384 cp > $o/pmc_traffic.log 2>&1; cp gpurun_out/pmc_traffic.json $o/pmc_traffic.json; tail -n 8 $o/pmc_traffic.json
tools/pmc_traffic.sh 16384 dd > $o/pmc_traffic_dd.log 2>&1; cp gpurun_out/pmc_traffic.json $o/pmc_traffic_dd.json; cp $o/pmc_traffic.json gpurun_out/pmc_traffic.json
# ... and of the channel decoder's kernels
tools/pmc_decoder.sh 16384 > $o/pmc_decoder.log 2>&1; cp gpurun_out/pmc_decoder.json $o/pmc_decoder.json
# SQ / LDS counters of every kernel of a bench run
tools/pmc_ofdm.sh $tag > $o/pmc_ofdm.log 2>&1; cp gpurun_out/pmc_ofdm_$tag/summary.md $o/pmc_ofdm_summary.md
# the decoder alone (per launch shape, and per kernel under the tracer), the one-frame path laid end to end, one ensemble per batch size
python3 tools/vit_time.py 16384 > $o/vit_time.txt 2>&1; cat $o/vit_time.txt
python3 tools/lane_shapes.py 16384 > $o/lane_shapes.txt 2>&1; cat $o/lane_shapes.txt
(cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --output-format csv -d $o/ls -o ls -- python3 $root/tools/lane_shapes.py 16384 > /dev/null 2>&1
 python3 $root/tools/lane_shapes_report.py $o/ls/ls_kernel_trace.csv > $o/lane_shapes_kernels.txt 2>&1; rm -rf $o/ls
 rocprofv3 --kernel-trace --output-format csv -d $o/ft -o ft -- python3 $root/tools/frame_timeline.py work 2>&1 | grep "host wall" > $o/frame_timeline.txt
 python3 $root/tools/frame_timeline.py report $o/ft/ft_kernel_trace.csv >> $o/frame_timeline.txt 2>&1; rm -rf $o/ft)
python3 tools/frame_timeline.py work >> $o/frame_timeline.txt 2>&1; cat $o/frame_timeline.txt
cat $o/lane_shapes_kernels.txt
(cd tools/ubench && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 row_step_cycles.hip -o row_step_cycles 2>/dev/null; ./row_step_cycles) > $o/row_step_cycles.txt 2>&1
python3 tools/frame_latency.py > $o/frame_latency.txt 2>&1; tail -n 3 $o/frame_latency.txt
python3 tools/single_time.py 1 4 64 256 1024 > $o/single_time.txt 2>&1; cat $o/single_time.txt     # one ensemble: each call of the step, per batch size
python3 tools/decoder_fuzz.py 400 1024 > $o/decoder_fuzz.txt 2>&1; tail -n 1 $o/decoder_fuzz.txt       # the two decoders against each other (and the oracle)
timeout 600 python3 tools/alloc_stress.py 100 100 > $o/alloc_stress.txt 2>&1; tail -n 2 $o/alloc_stress.txt
