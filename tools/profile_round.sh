#!/bin/bash
# One GPU-box visit for the evidence of a round: full bench line, rocprofv3 kernel-trace summary of the same
# command, PMC traffic passes, bench lines of the other workloads.  Everything lands under gpurun_out/round/.
set -u
R=gpurun_out/round; mkdir -p $R; export TMPDIR=/tmp
( timeout 900 python bench.py 2> $R/bench.err | tail -1 ) > $R/bench_c3.json
rm -rf gpurun_out/prof
( timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o trace -- python bench.py --steps 30 --warmup 3 --no-cpu-baseline --plain 2>&1 | tail -3 ) > $R/rocprof.log
DB=$(find gpurun_out/prof -name '*.db' | head -1)
if [ -n "$DB" ]; then python tools/rocpd_summary.py "$DB" $R/kernel_stats.md > /dev/null; fi
find gpurun_out/prof -name '*.db' -size +20M -delete
bash tools/pmc_round.sh > $R/pmc.log 2>&1
cp gpurun_out/pmc_summary.md $R/pmc_summary.md; cp gpurun_out/pmc_traffic_c3.json $R/pmc_traffic_c3.json 2>/dev/null
for wl in c2_4k_rgb_8b_rev53:8 c4_16k_gray_16b_rev53_tiled:16 c5_4k_444_10b_irv97_batch:16; do
  w=${wl%%:*}; c=${wl##*:}
  ( timeout 900 python bench.py --workload $w --container $c --steps 200 --no-cpu-baseline 2>> $R/bench.err | tail -1 ) > $R/bench_$w.json
done
tail -c 600 $R/bench_c3.json; echo; head -30 $R/kernel_stats.md; head -20 $R/pmc_summary.md
