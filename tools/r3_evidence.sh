#!/bin/bash
# round 3: the evidence of the final build in one box visit -> gpurun_out/r3ev/ (copied to profiles/r03_b_* afterwards)
set -u
R=gpurun_out/r3ev; mkdir -p $R; export TMPDIR=/tmp
rm -rf gpurun_out/prof
( timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o trace -- python bench.py --steps 30 --warmup 3 --no-cpu-baseline --plain 2>&1 | tail -3 ) > $R/rocprof.log
DB=$(find gpurun_out/prof -name '*.db' | head -1)
if [ -n "$DB" ]; then python tools/rocpd_summary.py "$DB" $R/kernel_stats.md > /dev/null; fi
find gpurun_out/prof -name '*.db' -size +20M -delete
timeout 400 bash tools/pmc_round.sh > $R/pmc.log 2>&1
cp gpurun_out/pmc_summary.md $R/pmc_summary.md; cp gpurun_out/pmc_traffic_c3.json $R/pmc_traffic_c3.json 2>/dev/null
timeout 400 bash tools/sq_round.sh > $R/sq_counters.txt 2>&1
cp gpurun_out/sq_counters.json $R/sq_counters.json 2>/dev/null
# the bench line last, with the counter files of THIS build in place (bench.py quotes nothing from a stale file)
cp gpurun_out/pmc_traffic_c3.json profiles/pmc_traffic.json 2>/dev/null; cp gpurun_out/sq_counters.json profiles/sq_counters.json 2>/dev/null
( timeout 600 python bench.py 2> $R/bench.err | tail -1 ) > $R/bench_c3.json
for wl in c2_4k_rgb_8b_rev53:8 c4_16k_gray_16b_rev53_tiled:16 c5_4k_444_10b_irv97_batch:16; do
  w=${wl%%:*}; c=${wl##*:}
  ( timeout 400 python bench.py --workload $w --container $c --steps 200 --no-cpu-baseline 2>> $R/bench2.err | tail -1 ) > $R/bench_$w.json
done
( OJPH_BENCH_BACKEND=gloo OJPH_BENCH_ONE_GPU=1 timeout 400 python bench.py --gpus 2 --steps 20 --no-cpu-baseline 2>> $R/bench2.err | tail -1 ) > $R/bench_2ranks_one_gpu.json
tail -c 700 $R/bench_c3.json; echo; head -24 $R/kernel_stats.md; head -16 $R/pmc_summary.md; tail -3 $R/bench.err
