#!/bin/bash
# PC sampling of the bench step (rocprofv3 beta feature): where the wavefronts of each kernel spend their time
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
M=${1:-stochastic}; U=${2:-cycles}; I=${3:-1048576}
rm -rf $R/gpurun_out/pcs
timeout 600 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $M --pc-sampling-unit $U --pc-sampling-interval $I --kernel-trace --output-format csv json -d $R/gpurun_out/pcs -- python $R/bench.py --steps 30 --warmup 2 --no-cpu-baseline --e2e-frames 0 > $R/gpurun_out/pcs.log 2>&1
tail -5 $R/gpurun_out/pcs.log | cut -c1-300
find $R/gpurun_out/pcs -type f | head -20
for f in $(find $R/gpurun_out/pcs -name "*pc_sampling*csv" | head -3); do echo == $f; head -5 $f; wc -l $f; done
