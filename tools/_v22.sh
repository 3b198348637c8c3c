#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 260 python tools/fuzz_narrow_gpu.py 200 31 2>&1 | grep -v amdgpu.ids | tail -8 ) > gpurun_out/r5_v22_narrow.txt; cat gpurun_out/r5_v22_narrow.txt
( timeout 320 python tools/fuzz_blocks_gpu.py 260 77 2>&1 | grep -v amdgpu.ids | tail -3 ) > gpurun_out/r5_v22_blocks.txt; cat gpurun_out/r5_v22_blocks.txt
( timeout 260 python tools/fuzz_part2_gpu.py 200 19 2>&1 | grep -v amdgpu.ids | tail -3 ) > gpurun_out/r5_v22_part2.txt; cat gpurun_out/r5_v22_part2.txt
( timeout 200 python tools/stress.py 60 2>&1 | grep -v amdgpu.ids | tail -6 ) > gpurun_out/r5_v22_stress.txt; cat gpurun_out/r5_v22_stress.txt
