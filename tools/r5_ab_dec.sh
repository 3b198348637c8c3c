#!/bin/bash
# round 5: A/B of block decoder builds inside one GPU visit.  usage: tools/r5_ab_dec.sh <variant> ...   (variants in openjph_amd/variants/lib_<v>.so; "orig" = the tree's build)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
cp openjph_amd/libojphgpu.so /tmp/lib_orig.so
run() {  # $1 = label, rest = env assignments
  local v=$1; shift
  if [ $v = orig ]; then cp /tmp/lib_orig.so openjph_amd/libojphgpu.so; else cp openjph_amd/variants/lib_$v.so openjph_amd/libojphgpu.so; fi
  env "$@" OJPH_BENCH_NOCHECK=1 timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --e2e-frames 0 2>/tmp/err.txt | tail -1 > /tmp/out.txt
  python - "$v" "$*" <<'PY'
import json,sys
try:
    d=json.loads(open('/tmp/out.txt').read()); k=d['kernels']
    f=[v['ms'] for n,v in k.items() if 'fused' in n]
    print('%-8s %-28s step %.4f enc %.4f dec %.4f | fused %s | inv %.4f | enc %s' % (sys.argv[1], sys.argv[2], d['ms_per_step'], d['config']['encode_ms'], d['config']['decode_ms'], ' '.join('%.4f'%x for x in f), k['dwt_inverse(all levels)']['ms'], ' '.join('%.3f' % v['ms'] for n,v in k.items() if 'encode' in n)))
except Exception as e:
    print(sys.argv[1], sys.argv[2], 'FAILED', e, open('/tmp/err.txt').read()[-300:])
PY
}
for rep in 1 2; do for v in "$@"; do run $v A=1; done; done
for v in "$@"; do run $v OJPHGPU_FUSED_DBG=1; run $v OJPHGPU_FUSED_DBG=2; done
cp /tmp/lib_orig.so openjph_amd/libojphgpu.so
