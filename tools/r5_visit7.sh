#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_codec.py tests/test_gpu_stages.py tests/test_gpu_contention.py tests/test_gpu_fullsize.py tests/test_gpu_damaged.py tests/test_gpu_pipeline.py tests/test_gpu_part2.py -q -m gpu -x 2>&1 | tail -30 ) > gpurun_out/r5_v7_tests.log
tail -4 gpurun_out/r5_v7_tests.log
bash tools/r5_ab_dec.sh orig base 2>&1 | tee gpurun_out/r5_v7_ab.txt
for wl in c6_4k_gray_32b_rev53 c7_4k_444_12b_atk97; do
  ( timeout 300 python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --e2e-frames 0 --no-strong 2> gpurun_out/r5_v7_$wl.err | tail -1 ) > gpurun_out/r5_v7_$wl.json
  python - $wl <<'PY'
import json,sys
try:
    d=json.loads(open('gpurun_out/r5_v7_%s.json' % sys.argv[1]).read())
    print(sys.argv[1], 'step', d['ms_per_step'], 'value', d['value'], 'enc', d['config']['encode_ms'], 'dec', d['config']['decode_ms'])
    for k,v in d['kernels'].items(): print('   ', k, v)
except Exception as e:
    print(sys.argv[1], 'FAILED', e, open('gpurun_out/r5_v7_%s.err' % sys.argv[1]).read()[-600:])
PY
done
