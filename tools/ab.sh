#!/bin/bash
# A/B of library builds inside one GPU-box visit: tools/ab.sh orig prev ...
cp openjph_amd/libojphgpu.so /tmp/lib_orig.so
for rep in 1 2; do
for v in "$@"; do
if [ $v = orig ]; then cp /tmp/lib_orig.so openjph_amd/libojphgpu.so; else cp openjph_amd/variants/lib_$v.so openjph_amd/libojphgpu.so; fi
OJPH_BENCH_NOCHECK=1 timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --e2e-frames 0 2>/tmp/err.txt | tail -1 > /tmp/out.txt
python -c "
import json,sys; d=json.loads(open('/tmp/out.txt').read()); k=d['kernels']
print('%-8s step %.4f enc %.4f dec %.4f | fwd %.4f L1 %.4f | inv %.4f L1 %.4f | prep %.3f s1 %.3f s2 %.3f | enc %s' % ('$v', d['ms_per_step'], d['config']['encode_ms'], d['config']['decode_ms'], k['dwt_forward(all levels)']['ms'], k['dwt_forward(level 1)']['ms'], k['dwt_inverse(all levels)']['ms'], k['dwt_inverse(level 1)']['ms'], k['ht_dec_prep']['ms'], k['ht_dec_step1']['ms'], k['ht_dec_step2']['ms'], ' '.join('%.3f' % v['ms'] for n,v in k.items() if 'encode' in n)))" 2>/dev/null || tail -3 /tmp/err.txt
done; done
cp /tmp/lib_orig.so openjph_amd/libojphgpu.so
