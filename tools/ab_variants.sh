#!/bin/bash
# A/B of library builds (tools/build_variant.py) inside one GPU-box visit, alternating, with the bench's own checks ON
# (round trip, failed blocks, the reference's digests after the timed loop):
#   tools/ab_variants.sh "orig v1 v2" [workload:container ...]   ->  gpurun_out/ab_variants.txt
set -u
VARS=$1; shift
WLS=${@:-c3_8k_444_12b_irv97:16}
O=gpurun_out/ab_variants.txt; : > $O
cp openjph_amd/libojphgpu.so /tmp/lib_orig.so
for wl in $WLS; do
  w=${wl%%:*}; c=${wl##*:}
  for rep in 1 2 3; do
    for v in $VARS; do
      if [ $v = orig ]; then cp /tmp/lib_orig.so openjph_amd/libojphgpu.so; else cp openjph_amd/variants/lib_$v.so openjph_amd/libojphgpu.so; fi
      timeout 300 python bench.py --workload $w --container $c --steps 300 --warmup 5 --no-cpu-baseline --plain 2>/tmp/err.txt | tail -1 > /tmp/out.txt
      python -c "
import json; d=json.loads(open('/tmp/out.txt').read()); c=d['config']
print('%-28s %-8s step %.4f enc %.4f dec %.4f' % ('$w', '$v', d['ms_per_step'], c['encode_ms'], c['decode_ms']), {k: v['ms'] for k, v in d['kernels'].items() if 'level 1' not in k}, 'verified' if d.get('verified_after_timing') else '')" >> $O 2>&1 || tail -3 /tmp/err.txt >> $O
    done
  done
done
cp /tmp/lib_orig.so openjph_amd/libojphgpu.so
cat $O
