#!/bin/bash
for mode in "X=0" "OJPHGPU_DWT_RP_INV=28" "OJPHGPU_DWT_RP_INV=40" "OJPHGPU_DWT_RP_INV=64" "OJPHGPU_DWT_RP_INV=12" "OJPHGPU_DWT_RP_FWD=28" "OJPHGPU_DWT_RP_FWD=12" "X=0"; do
  env $mode python bench.py --steps 60 --no-cpu-baseline --plain 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']
print('%-24s step %.4f enc %.4f dec %.4f | inv %.3f L1 %.4f | fwd %.3f L1 %.4f' % ('$mode', d['ms_per_step'], d['config']['encode_ms'], d['config']['decode_ms'], k['dwt_inverse(all levels)']['ms'], k['dwt_inverse(level 1)']['ms'], k['dwt_forward(all levels)']['ms'], k['dwt_forward(level 1)']['ms']))"
done
