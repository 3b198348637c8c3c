#!/bin/bash
cp openjph_amd/libojphgpu.so /tmp/lib_orig.so
for v in "$@"; do
if [ $v = orig ]; then cp /tmp/lib_orig.so openjph_amd/libojphgpu.so; else cp openjph_amd/csrc/_build/lib_$v.so openjph_amd/libojphgpu.so; fi
echo "== $v"
for f in 1 2; do
OJPH_BENCH_NOCHECK=1 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --frames $f 2>/tmp/err.txt | tail -1 > /tmp/out.txt
python -c "import json,sys; d=json.loads(open('/tmp/out.txt').read()); print(d['ms_per_step'], d['config']['encode_ms'], d['config']['decode_ms'], d['config'].get('roundtrip_max_abs_err')); [print('  ',k,v['ms']) for k,v in d['kernels'].items() if 'dec' in k]" 2>/dev/null || tail -5 /tmp/err.txt
done; done
cp /tmp/lib_orig.so openjph_amd/libojphgpu.so
