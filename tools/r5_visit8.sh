#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_codec.py tests/test_gpu_stages.py tests/test_gpu_contention.py tests/test_gpu_fullsize.py tests/test_gpu_damaged.py tests/test_gpu_pipeline.py -q -m gpu -x 2>&1 | tail -30 ) > gpurun_out/r5_v8_tests.log
tail -6 gpurun_out/r5_v8_tests.log
cp openjph_amd/libojphgpu.so /tmp/lib_orig.so
run() {
  local v=$1; shift
  if [ $v = orig ]; then cp /tmp/lib_orig.so openjph_amd/libojphgpu.so; else cp openjph_amd/variants/lib_$v.so openjph_amd/libojphgpu.so; fi
  env "$@" OJPH_BENCH_NOCHECK=1 timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --e2e-frames 0 --no-strong 2>/tmp/err.txt | tail -1 > /tmp/out.txt
  python - "$v" "$*" <<'PY'
import json,sys
try:
    d=json.loads(open('/tmp/out.txt').read()); k=d['kernels']
    f=[v['ms'] for n,v in k.items() if 'fused' in n]
    print('%-10s %-30s step %.4f enc %.4f dec %.4f | fused %s' % (sys.argv[1], sys.argv[2], d['ms_per_step'], d['config']['encode_ms'], d['config']['decode_ms'], ' '.join('%.4f'%x for x in f)))
except Exception as e:
    print(sys.argv[1], sys.argv[2], 'FAILED', e, open('/tmp/err.txt').read()[-300:])
PY
}
{
for rep in 1 2; do run orig A=1; run orig OJPHGPU_FUSED_PAIR=0; done
run orig OJPHGPU_FUSED_DBG=2
run orig OJPHGPU_FUSED_DBG=2 OJPHGPU_FUSED_PAIR=0
run orig OJPHGPU_FUSED_DBG=1
} 2>&1 | tee gpurun_out/r5_v8_ab.txt
cp /tmp/lib_orig.so openjph_amd/libojphgpu.so
