#!/bin/bash
# decoder A/B of library variants (no tests): tools/r3_dec_ab.sh orig mr2 ...
cp openjph_amd/libojphgpu.so /tmp/lib_dab_orig.so
for rep in 1 2; do for v in "$@"; do
  if [ $v = orig ]; then cp /tmp/lib_dab_orig.so openjph_amd/libojphgpu.so; else cp openjph_amd/variants/lib_$v.so openjph_amd/libojphgpu.so; fi
  python bench.py --steps 100 --no-cpu-baseline --plain 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']
print('%-8s step %.4f enc %.4f dec %.4f | prep %.3f s1 %.3f s2 %.3f | inv %.3f L1 %.3f | fwd %.3f L1 %.3f' % ('$v', d['ms_per_step'], d['config']['encode_ms'], d['config']['decode_ms'], k.get('ht_dec_prep',{'ms':0})['ms'], k.get('ht_dec_step1',{'ms':0})['ms'], k.get('ht_dec_step2',k.get('ht_dec_fused(step 1 + step 2)'))['ms'], k['dwt_inverse(all levels)']['ms'], k['dwt_inverse(level 1)']['ms'], k['dwt_forward(all levels)']['ms'], k['dwt_forward(level 1)']['ms']))"
done; done
cp /tmp/lib_dab_orig.so openjph_amd/libojphgpu.so
