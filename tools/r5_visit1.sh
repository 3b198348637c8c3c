#!/bin/bash
# round 5, first GPU visit: the new parity tests first, then the whole GPU suite, the Part-2 fuzzer, a bench line
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_damaged.py tests/test_gpu_stages.py tests/test_gpu_wide.py -q -m gpu -x 2>&1 | tail -40 ) > gpurun_out/r5_v1_new_tests.log
( timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -40 ) > gpurun_out/r5_v1_pytest_gpu.log
( timeout 200 python tools/fuzz_part2_gpu.py 90 5 2>&1 | tail -30 ) > gpurun_out/r5_v1_fuzz_part2.txt
( timeout 200 python tools/fuzz_blocks_gpu.py 40 710000 2>&1 | tail -30 ) > gpurun_out/r5_v1_fuzz_blocks.txt
( timeout 300 python bench.py 2> gpurun_out/r5_v1_bench.err | tail -1 ) > gpurun_out/r5_v1_bench.json
tail -5 gpurun_out/r5_v1_new_tests.log; tail -5 gpurun_out/r5_v1_pytest_gpu.log; tail -3 gpurun_out/r5_v1_fuzz_part2.txt; tail -3 gpurun_out/r5_v1_fuzz_blocks.txt; cut -c1-400 gpurun_out/r5_v1_bench.json
