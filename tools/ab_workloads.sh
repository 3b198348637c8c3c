#!/bin/bash
cp openjph_amd/libojphgpu.so /tmp/lib_orig.so
for rep in 1 2; do
for v in base rp20 rp32_4096; do
cp openjph_amd/csrc/_build/lib_$v.so openjph_amd/libojphgpu.so
for w in "c2_4k_rgb_8b_rev53" "c4_16k_gray_16b_rev53_tiled" "c5_4k_444_10b_irv97_batch --frames 8" "c5_4k_444_10b_irv97_batch --frames 1"; do
OJPH_BENCH_NOCHECK=1 timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']
print('%-10s %-44s step %.4f | fwd %.4f L1 %.4f | inv %.4f L1 %.4f' % ('$v', '$w', d['ms_per_step'], k['dwt_forward(all levels)']['ms'], k['dwt_forward(level 1)']['ms'], k['dwt_inverse(all levels)']['ms'], k['dwt_inverse(level 1)']['ms']))"
done; done; done
cp /tmp/lib_orig.so openjph_amd/libojphgpu.so
