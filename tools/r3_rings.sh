#!/bin/bash
# round 3: the fused launch's workers with a ring per block / all records of a slice requested at once -- parity first, then A/B
timeout 400 python -m pytest tests/test_gpu_codec.py tests/test_gpu_stages.py -m gpu -x -q --timeout 120 2>&1 | tail -4
timeout 120 python tools/dbg_fused.py 2>&1 | tail -3
for mode in "X=0" "OJPHGPU_FUSED_RINGS=1" "X=0" "OJPHGPU_FUSED_RINGS=1"; do
  env $mode python bench.py --steps 100 --no-cpu-baseline --plain --no-strong 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']
print('%-22s step %.4f enc %.4f dec %.4f | fused %.4f | inv %.3f' % ('$mode', d['ms_per_step'], d['config']['encode_ms'], d['config']['decode_ms'], k.get('ht_dec_step2',k.get('ht_dec_fused(step 1 + step 2)'))['ms'], k['dwt_inverse(all levels)']['ms']))"
done
env python bench.py --workload c2_4k_rgb_8b_rev53 --container 8 --steps 200 --no-cpu-baseline --plain --no-strong 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c2', d['ms_per_step'], d['config']['encode_ms'], d['config']['decode_ms'])"
