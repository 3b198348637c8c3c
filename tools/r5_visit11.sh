#!/bin/bash
# round 5, visit 11: encoder prologue variants, general batches merged per kernel (c7), duplex pipes by copy mode
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_part2.py tests/test_gpu_wide.py tests/test_gpu_codec.py -q -m gpu -x 2>&1 | tail -8 ) > gpurun_out/r5_v11_tests.log; tail -3 gpurun_out/r5_v11_tests.log
cp openjph_amd/libojphgpu.so /tmp/lib_orig.so
for v in orig pro0 pro1 orig pro0 pro1; do
  if [ $v = orig ]; then cp /tmp/lib_orig.so openjph_amd/libojphgpu.so; else cp openjph_amd/variants/lib_$v.so openjph_amd/libojphgpu.so; fi
  echo "== $v"; timeout 300 python tools/block_sizes.py 64x64 32x32 2>&1 | grep "^block" | cut -c1-75
done | tee gpurun_out/r5_v11_prologue_ab.txt
cp /tmp/lib_orig.so openjph_amd/libojphgpu.so
for wl in c7_4k_444_12b_atk97 c6_4k_gray_32b_rev53; do
  timeout 300 python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --e2e-frames 0 --no-strong 2>/tmp/err.txt | tail -1 > gpurun_out/r5_v11_$wl.json
  python - $wl <<'PY'
import json,sys
try:
    d=json.loads(open('gpurun_out/r5_v11_%s.json' % sys.argv[1]).read()); k=d['kernels']
    print(sys.argv[1], 'step', d['ms_per_step'], 'value', d['value'], 'enc', d['config']['encode_ms'], 'dec', d['config']['decode_ms'], '| dwt fwd', k['dwt_forward(all levels)'], 'inv', k['dwt_inverse(all levels)'])
except Exception as e:
    print(sys.argv[1], 'FAILED', e, open('/tmp/err.txt').read()[-600:])
PY
done
( timeout 600 python tools/e2e_duplex.py --frames 48 2>&1 | grep -v amdgpu.ids ) | tee gpurun_out/r5_v11_duplex.txt
( timeout 300 python tools/e2e_duplex.py --frames 48 --packed 12 --modes 00,22,10 2>&1 | grep -v amdgpu.ids ) | tee gpurun_out/r5_v11_duplex_packed.txt
