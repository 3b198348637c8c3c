#!/bin/bash
# block-encoder iteration: parity tests of the stages and codecs that use it, then an A/B of library builds (encoder only)
set -u
R=gpurun_out/r3enc; mkdir -p $R; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_codec.py -m gpu -x -q --timeout 120 2>&1 | tail -15 ) > $R/pytest.txt
tail -6 $R/pytest.txt
( timeout 600 python tools/enc_only.py "$@" 2>&1 ) | tee $R/enc_only.txt
