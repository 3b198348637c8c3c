#!/bin/bash
for mode in "OJPHGPU_FUSED_DBG=0" "OJPHGPU_FUSED_DBG=2" "OJPHGPU_FUSED_DBG=1"; do
  env OJPH_BENCH_NOCHECK=1 $mode python bench.py --steps 60 --no-cpu-baseline --plain 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']
print('%-22s step %.4f enc %.4f dec %.4f | fused %.3f | inv %.3f' % ('$mode', d['ms_per_step'], d['config']['encode_ms'], d['config']['decode_ms'], k.get('ht_dec_step2',k.get('ht_dec_fused(step 1 + step 2)'))['ms'], k['dwt_inverse(all levels)']['ms']))"
done
