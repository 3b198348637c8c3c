#!/bin/bash
# block-decoder iteration: parity tests of the stages and codecs, then the step's per-launch times with and without the prep launch
set -u
R=gpurun_out/r3dec; mkdir -p $R; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_codec.py -m gpu -x -q 2>&1 | tail -15 ) > $R/pytest.txt
tail -6 $R/pytest.txt
for prep in 1 0 1 0; do
  OJPHGPU_DEC_PREP=$prep python bench.py --steps 100 --no-cpu-baseline --plain 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']
print('prep=$prep step %.4f enc %.4f dec %.4f | prep %.3f s1 %.3f s2 %.3f | inv %.3f' % (d['ms_per_step'], d['config']['encode_ms'], d['config']['decode_ms'], k['ht_dec_prep']['ms'], k['ht_dec_step1']['ms'], k['ht_dec_step2']['ms'], k['dwt_inverse(all levels)']['ms']))"
done | tee $R/ab.txt
