#!/bin/bash
for mode in "X=0" "OJPHGPU_ENC_SIDE_PRIO=low" "OJPHGPU_ENC_SIDE_PRIO=high" "X=0" "OJPHGPU_ENC_SIDE_PRIO=low"; do
  env $mode python bench.py --steps 100 --no-cpu-baseline --plain 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']
print('%-28s step %.4f enc %.4f dec %.4f | enc launches %s | fwd %.3f' % ('$mode', d['ms_per_step'], d['config']['encode_ms'], d['config']['decode_ms'], ' '.join('%.3f' % v['ms'] for n,v in k.items() if 'encode' in n), k['dwt_forward(all levels)']['ms']))"
done
