#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_wide.py tests/test_gpu_part2.py tests/test_gpu_stages.py tests/test_gpu_codec.py tests/test_gpu_fullsize.py -q -m gpu -x 2>&1 | tail -30 ) > gpurun_out/r5_v9_tests.log
tail -6 gpurun_out/r5_v9_tests.log
( timeout 200 python tools/fuzz_part2_gpu.py 60 11 2>&1 | tail -5 ) > gpurun_out/r5_v9_fuzz_part2.txt; tail -2 gpurun_out/r5_v9_fuzz_part2.txt
for wl in c6_4k_gray_32b_rev53 c7_4k_444_12b_atk97; do
  for mode in 0 1; do
  ( OJPHGPU_LIFT_ELEMENTWISE=$mode timeout 300 python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --e2e-frames 0 --no-strong 2> gpurun_out/r5_v9_$wl.err | tail -1 ) > gpurun_out/r5_v9_${wl}_$mode.json
  python - $wl $mode <<'PY'
import json,sys
try:
    d=json.loads(open('gpurun_out/r5_v9_%s_%s.json' % (sys.argv[1], sys.argv[2])).read())
    k=d['kernels']
    print(sys.argv[1], 'elementwise' if sys.argv[2]=='1' else 'pipeline', 'step', d['ms_per_step'], 'value', d['value'], 'enc', d['config']['encode_ms'], 'dec', d['config']['decode_ms'],
          '| dwt fwd', k['dwt_forward(all levels)'], 'inv', k['dwt_inverse(all levels)'])
except Exception as e:
    print(sys.argv[1], 'FAILED', e, open('gpurun_out/r5_v9_%s.err' % sys.argv[1]).read()[-600:])
PY
  done
done
# decoder tail-split variants
cp openjph_amd/libojphgpu.so /tmp/lib_orig.so
for rep in 1 2; do for v in orig tail0 tail1; do
  if [ $v = orig ]; then cp /tmp/lib_orig.so openjph_amd/libojphgpu.so; else cp openjph_amd/variants/lib_$v.so openjph_amd/libojphgpu.so; fi
  OJPH_BENCH_NOCHECK=1 timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --e2e-frames 0 --no-strong 2>/tmp/err.txt | tail -1 > /tmp/out.txt
  python - $v <<'PY'
import json,sys
d=json.loads(open('/tmp/out.txt').read()); k=d['kernels']
print('%-8s step %.4f enc %.4f dec %.4f | fused %s' % (sys.argv[1], d['ms_per_step'], d['config']['encode_ms'], d['config']['decode_ms'], ' '.join('%.4f'%v['ms'] for n,v in k.items() if 'fused' in n)))
PY
done; done
cp /tmp/lib_orig.so openjph_amd/libojphgpu.so
