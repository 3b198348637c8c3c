#!/bin/bash
# throughput of frame batches (BASELINE config 5: independent 4K frames) and of 8K frames, by batch size
for wl in "c5_4k_444_10b_irv97_batch 1 4 8 16" "c3_8k_444_12b_irv97 1 2 4"; do
  set -- $wl; w=$1; shift
  for f in "$@"; do
    timeout 300 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --frames $f 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']
print('%-28s frames %2d  %8.0f Msamples/s  step %.3f ms | prep %.3f s1 %.3f s2 %.3f' % ('$w', d['config']['frames_per_step'], d['value'], d['ms_per_step'], k['ht_dec_prep']['ms'], k['ht_dec_step1']['ms'], k['ht_dec_step2']['ms']))"
  done
done
