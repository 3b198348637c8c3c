#!/bin/bash
# A/B of library builds for the fused decoder launch inside one GPU-box visit: whole launch, chains alone (dbg 1), workers
# alone over the previous run's records (dbg 2).   tools/r4_fused_ab.sh orig old ...
cp openjph_amd/libojphgpu.so /tmp/lib_orig.so
for rep in 1 2; do
for v in "$@"; do
if [ $v = orig ]; then cp /tmp/lib_orig.so openjph_amd/libojphgpu.so; else cp openjph_amd/variants/lib_$v.so openjph_amd/libojphgpu.so; fi
for mode in ${MODES:-0 1 2}; do
  env OJPH_BENCH_NOCHECK=1 OJPHGPU_FUSED_DBG=$mode timeout 200 python bench.py --steps 60 --no-cpu-baseline --plain ${WORKLOAD:+--workload $WORKLOAD} 2>/tmp/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('%-8s dbg $mode  step %.4f enc %.4f dec %.4f | fused %.4f | inv %.3f' % ('$v', d['ms_per_step'], d['config']['encode_ms'], d['config']['decode_ms'], k.get('ht_dec_step2',k.get('ht_dec_fused(step 1 + step 2)'))['ms'], k['dwt_inverse(all levels)']['ms']))" 2>/dev/null || tail -3 /tmp/err.txt
done; done; done
cp /tmp/lib_orig.so openjph_amd/libojphgpu.so
