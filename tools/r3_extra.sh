#!/bin/bash
# round 3: code-block size sweep, fused launch against the separate launches
mkdir -p gpurun_out/r3x
OJPHGPU_DEC_FUSED=0 timeout 300 python tools/block_sizes.py > gpurun_out/r3x/block_sizes_separate.txt 2>&1; tail -5 gpurun_out/r3x/block_sizes_separate.txt
timeout 300 python tools/block_sizes.py > gpurun_out/r3x/block_sizes.txt 2>&1; tail -5 gpurun_out/r3x/block_sizes.txt
