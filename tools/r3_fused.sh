#!/bin/bash
timeout 300 python -m pytest tests/test_gpu_codec.py -m gpu -x -q --timeout 120 2>&1 | tail -3
for mode in "OJPHGPU_FUSED_SHAPE=1" "OJPHGPU_FUSED_SHAPE=0" "OJPHGPU_FUSED_SHAPE=1" "OJPHGPU_FUSED_SHAPE=0" "OJPHGPU_DEC_FUSED=0"; do
  env $mode python bench.py --steps 60 --no-cpu-baseline --plain 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']
print('%-22s step %.4f enc %.4f dec %.4f | prep %.3f s1 %.3f s2 %.3f | inv %.3f' % ('$mode', d['ms_per_step'], d['config']['encode_ms'], d['config']['decode_ms'], k.get('ht_dec_prep',{'ms':0})['ms'], k.get('ht_dec_step1',{'ms':0})['ms'], k.get('ht_dec_step2',k.get('ht_dec_fused(step 1 + step 2)'))['ms'], k['dwt_inverse(all levels)']['ms']))"
done
