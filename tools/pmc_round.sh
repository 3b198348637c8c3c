#!/bin/bash
# HBM traffic counters, each in its own pass (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2).
# Only --pmc + --kernel-trace: never combined with sys/hip/hsa tracing.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$C
  ( timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_$C -o pmc -- \
      python bench.py --steps 3 --warmup 1 --no-cpu-baseline --plain --calibrate 2>&1 | tail -2 ) > gpurun_out/pmc_$C.log
  find gpurun_out/pmc_$C -type f | head -5 >> gpurun_out/pmc_$C.log
done
python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_traffic_c3.json > gpurun_out/pmc_summary.md 2> gpurun_out/pmc_summary.err
cat gpurun_out/pmc_summary.md | head -30
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -type f -size +8M -delete
