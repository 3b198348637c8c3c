#!/bin/bash
# round 3, first box visit: the new tests, the VALU-issue probe, a baseline bench line with the strong-scaling part
set -u
R=gpurun_out/r3v1; mkdir -p $R; export TMPDIR=/tmp
( timeout 120 tools/micro/valu_issue 200 ) > $R/valu_issue.txt 2>&1
( timeout 900 python -m pytest tests/test_bench_contract.py tests/test_gpu_codec.py tests/test_gpu_pipeline.py -m gpu -x -q -k "two_ranks or live_bench or foreign or fuzz_seed or pixel_interleaved or bit_packed" 2>&1 | tail -15 ) > $R/pytest_new.txt
( timeout 600 python bench.py --steps 300 2> $R/bench.err | tail -1 ) > $R/bench_c3.json
tail -5 $R/pytest_new.txt; head -50 $R/valu_issue.txt; tail -c 2500 $R/bench_c3.json; tail -3 $R/bench.err
