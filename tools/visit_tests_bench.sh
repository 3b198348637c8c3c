#!/bin/bash
# One GPU-box visit: the default bench line, then the whole GPU suite exactly as the driver runs it (-x), LAST.
set -u
O=gpurun_out/v1; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python bench.py 2> $O/bench.err | tail -1 ) > $O/bench_c3.json
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 ) > $O/pytest_gpu.log
tail -c 1500 $O/bench_c3.json; echo; tail -5 $O/bench.err; tail -12 $O/pytest_gpu.log
