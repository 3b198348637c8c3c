#!/bin/bash
# The evidence of a build in ONE GPU-box visit -> gpurun_out/ev/ (copy what is to be judged to profiles/rNN_<tag>_*):
#   1. counters FIRST -- kernel trace of the bench step (-> kernel_stats.md), HBM traffic counters of every bench workload
#      (separate FETCH_SIZE / WRITE_SIZE passes, --pmc + --kernel-trace only), SQ counters of C3 -- and the two counter files
#      put in place (profiles/pmc_traffic.json, profiles/sq_counters.json: stamped with the digest of the kernel sources);
#   2. the bench lines of every workload, which quote those files;
#   3. the GPU suite exactly as the driver runs it (-m gpu -x), LAST.  Nothing but its log is committed after this script:
#      the tree that was tested is the tree that ships (round 5 shipped a tree its own suite had never seen).
# Usage: tools/evidence.sh [quick]     (quick: C3 only, no fuzz runs)
set -u
R=gpurun_out/ev; mkdir -p $R; export TMPDIR=/tmp
QUICK=${1:-}
rm -rf gpurun_out/prof
( timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o trace -- python bench.py --steps 30 --warmup 3 --no-cpu-baseline --plain 2>&1 | tail -3 ) > $R/rocprof.log
DB=$(find gpurun_out/prof -name '*.db' | head -1)
if [ -n "$DB" ]; then python tools/rocpd_summary.py "$DB" $R/kernel_stats.md > /dev/null; fi
find gpurun_out/prof -name '*.db' -size +20M -delete
WLS="c3_8k_444_12b_irv97:16"
[ -z "$QUICK" ] && WLS="$WLS c2_4k_rgb_8b_rev53:8 c4_16k_gray_16b_rev53_tiled:16 c5_4k_444_10b_irv97_batch:16"
for wl in $WLS; do
  w=${wl%%:*}; c=${wl##*:}
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_$C
    ( OJPH_BENCH_NOCHECK=1 timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_$C -o pmc -- \
        python bench.py --workload $w --container $c --steps 3 --warmup 1 --no-cpu-baseline --plain --calibrate 2>&1 | tail -2 ) > $R/pmc_${w}_$C.log
  done
  python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE $R/pmc_traffic_$w.json $w > $R/pmc_summary_$w.md 2>> $R/pmc_summary.err
  find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -type f -size +8M -delete
done
python - <<'PY'
import glob, json
out = {}
for f in sorted(glob.glob("gpurun_out/ev/pmc_traffic_c*.json")):
    j = json.load(open(f))
    out.setdefault("_kernels_sha256", j["_kernels_sha256"])
    assert out["_kernels_sha256"] == j["_kernels_sha256"]
    out.update({k: v for k, v in j.items() if not k.startswith("_")})
json.dump(out, open("gpurun_out/ev/pmc_traffic.json", "w"), indent=1)
PY
rm -f gpurun_out/sq_counters.json
timeout 400 bash tools/sq_round.sh > $R/sq_counters.txt 2>&1
cp gpurun_out/sq_counters.json $R/sq_counters.json 2>/dev/null
cp $R/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null; cp gpurun_out/sq_counters.json profiles/sq_counters.json 2>/dev/null
# 2. the bench lines, with the counter files of THIS build in place (bench.py quotes nothing from a stale file)
( timeout 900 python bench.py 2> $R/bench.err | tail -1 ) > $R/bench_c3.json
if [ -z "$QUICK" ]; then
  for wl in c2_4k_rgb_8b_rev53:8 c4_16k_gray_16b_rev53_tiled:16 c5_4k_444_10b_irv97_batch:16; do
    w=${wl%%:*}; c=${wl##*:}
    ( timeout 400 python bench.py --workload $w --container $c --steps 200 --no-cpu-baseline 2>> $R/bench2.err | tail -1 ) > $R/bench_$w.json
  done
  for wl in c6_4k_gray_32b_rev53:32 c7_4k_444_12b_atk97:16; do
    w=${wl%%:*}; c=${wl##*:}
    ( timeout 500 python bench.py --workload $w --container $c --steps 200 2>> $R/bench2.err | tail -1 ) > $R/bench_$w.json
  done
  ( OJPH_BENCH_BACKEND=gloo OJPH_BENCH_ONE_GPU=1 timeout 400 python bench.py --gpus 2 --steps 20 --no-cpu-baseline 2>> $R/bench2.err | tail -1 ) > $R/bench_2ranks_one_gpu.json
  ( timeout 300 python tools/block_sizes.py 2>&1 | grep "^block" ) > $R/block_sizes.txt
  ( SWEEP_STAGES=1 timeout 300 python tools/qstep_sweep.py 2>&1 | grep "qstep\|stages\|alone" ) > $R/rate_sweep.txt
  ( timeout 100 python tools/fuzz_blocks_gpu.py 45 5 2>&1 | tail -2 ) > $R/fuzz_blocks.txt
  ( timeout 100 python tools/fuzz_part2_gpu.py 45 7 2>&1 | tail -2 ) > $R/fuzz_part2.txt
  bash tools/ab_env.sh OJPHGPU_DWT_TRIP 1 2 c3_8k_444_12b_irv97:16 c2_4k_rgb_8b_rev53:8 c4_16k_gray_16b_rev53_tiled:16 c5_4k_444_10b_irv97_batch:16 > /dev/null 2>&1; cp gpurun_out/ab_env.txt $R/ab_dwt_trip.txt
fi
# 3. the GPU suite, last
( echo "HEAD $(git rev-parse --short HEAD 2>/dev/null || echo '(snapshot)') kernels $(python -c 'from openjph_amd.build import kernel_sources_digest as d; print(d()[:12])')"; \
  timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -25 ) > $R/gpu_tests.txt
tail -c 1200 $R/bench_c3.json; echo; head -16 $R/kernel_stats.md; head -8 $R/pmc_summary_c3_8k_444_12b_irv97.md; tail -3 $R/bench.err; tail -4 $R/gpu_tests.txt
