#!/bin/bash
# round 5, visit 10: link probe (duplex pipes), block sizes with the 16-byte table copy, C4 schedule with and without the overlap,
# inverse-DWT chunk heights, and the two-rank bench test with both gathers checked
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 120 tools/micro/link_duplex 256 2>&1 ) > gpurun_out/r5_v10_link_duplex.txt; cat gpurun_out/r5_v10_link_duplex.txt
( timeout 600 python -m pytest tests/test_gpu_stages.py -q -m gpu -x 2>&1 | tail -4 ) > gpurun_out/r5_v10_tests.log; tail -3 gpurun_out/r5_v10_tests.log
cp openjph_amd/libojphgpu.so /tmp/lib_orig.so
for v in orig tabold orig tabold; do
  if [ $v = orig ]; then cp /tmp/lib_orig.so openjph_amd/libojphgpu.so; else cp openjph_amd/variants/lib_$v.so openjph_amd/libojphgpu.so; fi
  echo "== $v"; timeout 300 python tools/block_sizes.py 64x64 32x32 2>&1 | grep "^block"
done | tee gpurun_out/r5_v10_block_sizes_ab.txt
cp /tmp/lib_orig.so openjph_amd/libojphgpu.so
timeout 300 python tools/block_sizes.py 2>&1 | grep "^block" | tee gpurun_out/r5_v10_block_sizes.txt
line() { python - "$@" <<'PY'
import json,sys
try:
    d=json.loads(open('/tmp/out.txt').read()); k=d['kernels']
    print('%-34s step %.4f enc %.4f dec %.4f |' % (' '.join(sys.argv[1:]), d['ms_per_step'], d['config']['encode_ms'], d['config']['decode_ms']),
          ' '.join('%s %.4f' % (n.split('(')[0][:14], v['ms']) for n,v in k.items() if 'inverse' in n or 'fused' in n or 'step' in n))
except Exception as e:
    print(' '.join(sys.argv[1:]), 'FAILED', e, open('/tmp/err.txt').read()[-400:])
PY
}
for wl in c4_16k_gray_16b_rev53_tiled c3_8k_444_12b_irv97; do
  for env in A=1 OJPHGPU_NO_OVERLAP=1 OJPHGPU_DWT_RP_INV=28 OJPHGPU_DWT_RP_INV=36; do
    env $env OJPH_BENCH_NOCHECK=1 timeout 300 python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --e2e-frames 0 --no-strong 2>/tmp/err.txt | tail -1 > /tmp/out.txt
    line $wl $env
  done
done | tee gpurun_out/r5_v10_schedule.txt
( timeout 900 python -m pytest tests/test_bench_contract.py -q -m gpu -x -k two_ranks 2>&1 | tail -15 ) > gpurun_out/r5_v10_tworanks.log; tail -5 gpurun_out/r5_v10_tworanks.log
