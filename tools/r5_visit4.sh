#!/bin/bash
# round 5: where does step 2's time go?  ablation builds (S2_ABL, kernels_ht_dec.hip), workers alone (OJPHGPU_FUSED_DBG=2) and the whole launch
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
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
for v in orig s2abl1 s2abl2 s2abl4 s2abl8 s2abl16 s2abl31; do run $v OJPHGPU_FUSED_DBG=2; done
for v in orig s2abl1 s2abl2 s2abl4 s2abl8 s2abl16 s2abl31; do run $v A=1; done
run orig OJPHGPU_FUSED_DBG=1
} 2>&1 | tee gpurun_out/r5_v4_abl.txt
cp /tmp/lib_orig.so openjph_amd/libojphgpu.so
