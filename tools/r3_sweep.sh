#!/bin/bash
# round 3: randomized parity sweep (GPU codec against the oracle pipeline, bytes and samples) on seeds the suite does not
# run: the block decoder's default schedule, the one launch wherever it is able to, and its one-ring worker form
mkdir -p gpurun_out/r3s
( timeout 200 python tools/gpu_sweep.py 7000 8500 2>&1 | tail -2 ) | tee gpurun_out/r3s/sweep_default.txt
( OJPHGPU_DEC_FUSED=2 timeout 200 python tools/gpu_sweep.py 8500 10000 2>&1 | tail -2 ) | tee gpurun_out/r3s/sweep_fused_always.txt
( OJPHGPU_DEC_FUSED=2 OJPHGPU_FUSED_RINGS=1 timeout 200 python tools/gpu_sweep.py 10000 11000 2>&1 | tail -2 ) | tee gpurun_out/r3s/sweep_fused_one_ring.txt
