#!/bin/bash
# One GPU-box visit: parity tests, bench, rocprof kernel trace.  Outputs under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/pytest_gpu.log
( timeout 600 python bench.py 2> gpurun_out/bench.err | tail -1 ) > gpurun_out/bench.json
rm -rf gpurun_out/prof
( timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o trace -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --e2e-frames 0 2>&1 | tail -3 ) > gpurun_out/rocprof.log
ls -R gpurun_out/prof | head -20 >> gpurun_out/rocprof.log
DB=$(find gpurun_out/prof -name '*.db' | head -1)
if [ -n "$DB" ]; then python tools/rocpd_summary.py "$DB" gpurun_out/kernel_stats.md > /dev/null; fi
find gpurun_out/prof -name '*.db' -size +20M -delete
cat gpurun_out/pytest_gpu.log | tail -5; cat gpurun_out/bench.json
