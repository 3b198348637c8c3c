#!/bin/bash
# round 5, visit 13: the 4K RGB frame (C2): chunk heights of the colour-fused top DWT level, fusion off, decoder launch shape
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
line() { python - "$@" <<'PY'
import json,sys
try:
    d=json.loads(open('/tmp/out.txt').read()); k=d['kernels']
    print('%-40s step %.4f enc %.4f dec %.4f |' % (' '.join(sys.argv[1:]), d['ms_per_step'], d['config']['encode_ms'], d['config']['decode_ms']),
          ' '.join('%s %.4f' % (n.replace('dwt_','').replace('(all levels)','*').replace('(level 1)','1')[:12], v['ms']) for n,v in k.items() if 'dwt' in n or 'fused' in n or 'convert' in n))
except Exception as e:
    print(' '.join(sys.argv[1:]), 'FAILED', e, open('/tmp/err.txt').read()[-400:])
PY
}
for rep in 1 2; do
for env in A=1 OJPHGPU_DWT_RP_COLOUR=4 OJPHGPU_DWT_RP_COLOUR=6 OJPHGPU_DWT_RP_COLOUR=12 OJPHGPU_DWT_RP_COLOUR=16 OJPHGPU_DWT_RP_COLOUR=24 OJPHGPU_NO_COLOUR_FUSION=1 OJPHGPU_FUSED_SHAPE=0 OJPHGPU_DEC_FUSED=0; do
  env $env OJPH_BENCH_NOCHECK=1 timeout 200 python bench.py --workload c2_4k_rgb_8b_rev53 --container 8 --steps 100 --warmup 5 --no-cpu-baseline --e2e-frames 0 --no-strong 2>/tmp/err.txt | tail -1 > /tmp/out.txt
  line c2 $env
done; done | tee gpurun_out/r5_v13_c2.txt
