#!/bin/bash
# where does the one-launch block decoder pay now?  decode time of the other workloads with OJPHGPU_DEC_FUSED = 0 (never), 1 (the
# library's rule), 2 (wherever it can)
for w in "c2_4k_rgb_8b_rev53" "c4_16k_gray_16b_rev53_tiled" "c5_4k_444_10b_irv97_batch --frames 8" "c5_4k_444_10b_irv97_batch --frames 4" "c5_4k_444_10b_irv97_batch --frames 1"; do
for m in 0 1 2; do
OJPHGPU_DEC_FUSED=$m OJPH_BENCH_NOCHECK=1 timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --plain 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']
print('fused=%s %-44s step %.4f enc %.4f dec %.4f | %s' % ('$m', '$w', d['ms_per_step'], d['config']['encode_ms'], d['config']['decode_ms'], ' '.join('%s %.3f' % (n[:14], v['ms']) for n, v in k.items() if 'dec' in n)))"
done; done
