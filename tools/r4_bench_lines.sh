#!/bin/bash
# the bench lines of the kept build again (after bench.py gained fields; the counters of profiles/ are of the same kernels) -> gpurun_out/r4ev/
R=gpurun_out/r4ev; mkdir -p $R
( timeout 600 python bench.py 2> $R/bench.err | tail -1 ) > $R/bench_c3.json
for wl in c2_4k_rgb_8b_rev53:8 c4_16k_gray_16b_rev53_tiled:16 c5_4k_444_10b_irv97_batch:16; do
  w=${wl%%:*}; c=${wl##*:}
  ( timeout 400 python bench.py --workload $w --container $c --steps 200 --no-cpu-baseline 2>> $R/bench2.err | tail -1 ) > $R/bench_$w.json
done
( OJPH_BENCH_BACKEND=gloo OJPH_BENCH_ONE_GPU=1 timeout 400 python bench.py --gpus 2 --steps 20 --no-cpu-baseline 2>> $R/bench2.err | tail -1 ) > $R/bench_2ranks_one_gpu.json
( timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 2>> $R/bench2.err | tail -1 ) > $R/bench_c3_driver_flags.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4ev/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
        print(f.split("/")[-1], d["value"], d["ms_per_step"], r["bound"], r["frac"], r.get("traffic_ratio"), r.get("frac_valu_issue"))
    except Exception as e:
        print(f, "ERR", e)
PY
