#!/bin/bash
# round 3: the fused launch's workers (ring per block, a slice's loads in flight while the one before is decoded) -- parity first, then A/B
timeout 400 python -m pytest tests/test_gpu_codec.py tests/test_gpu_stages.py -m gpu -x -q --timeout 120 2>&1 | tail -4
timeout 120 python tools/dbg_fused.py 2>&1 | tail -3
run() {
  env $1 python bench.py --steps 100 --no-cpu-baseline --plain --no-strong 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']
print('%-22s step %.4f enc %.4f dec %.4f | fused %.4f | inv %.3f' % ('$2', d['ms_per_step'], d['config']['encode_ms'], d['config']['decode_ms'], k.get('ht_dec_step2',k.get('ht_dec_fused(step 1 + step 2)'))['ms'], k['dwt_inverse(all levels)']['ms']))"
}
cp openjph_amd/libojphgpu.so /tmp/lib_orig.so
run X=0 default; run X=0 default
for v in "$@"; do
  cp openjph_amd/variants/lib_$v.so openjph_amd/libojphgpu.so
  timeout 200 python -m pytest tests/test_gpu_codec.py -m gpu -x -q --timeout 120 -k "decod or foreign or corrupt" 2>&1 | tail -1
  run X=0 $v
done
cp /tmp/lib_orig.so openjph_amd/libojphgpu.so
env python bench.py --workload c2_4k_rgb_8b_rev53 --container 8 --steps 200 --no-cpu-baseline --plain --no-strong 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c2', d['ms_per_step'], d['config']['encode_ms'], d['config']['decode_ms'])"
