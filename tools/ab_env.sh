#!/bin/bash
# A/B of ONE build under an environment switch, alternating, in one box visit:
#   tools/ab_env.sh VAR A B [workload:container ...]   -> gpurun_out/ab_env.txt (ms_per_step, encode_ms, decode_ms per run)
set -u
VAR=$1; A=$2; B=$3; shift 3
WLS=${@:-c3_8k_444_12b_irv97:16}
O=gpurun_out/ab_env.txt; : > $O
for wl in $WLS; do
  w=${wl%%:*}; c=${wl##*:}
  for round in 1 2 3; do
    for v in $A $B; do
      env $VAR=$v timeout 300 python bench.py --workload $w --container $c --steps 300 --warmup 5 --no-cpu-baseline --plain 2>/dev/null | tail -1 | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$w $VAR=$v step %.4f enc %.4f dec %.4f' % (d['ms_per_step'], c['encode_ms'], c['decode_ms']), {k: v['ms'] for k, v in d['kernels'].items()})" >> $O 2>&1
    done
  done
done
cat $O
