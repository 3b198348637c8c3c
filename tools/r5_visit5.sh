#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_codec.py tests/test_gpu_stages.py tests/test_gpu_contention.py tests/test_gpu_fullsize.py tests/test_gpu_damaged.py tests/test_gpu_pipeline.py tests/test_cli.py -q -m gpu -x 2>&1 | tail -30 ) > gpurun_out/r5_v5_tests.log
tail -4 gpurun_out/r5_v5_tests.log
bash tools/r5_ab_dec.sh orig base 2>&1 | tee gpurun_out/r5_v5_ab.txt
