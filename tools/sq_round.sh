#!/bin/bash
# SQ counters (instruction mix + stall composition), one pass of 8 SQ counters, --pmc + --kernel-trace only.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/sq1 gpurun_out/sq2
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT --kernel-trace --output-format csv -d gpurun_out/sq1 -o sq -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --e2e-frames 0 > gpurun_out/sq1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/sq2 -o sq -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --e2e-frames 0 > gpurun_out/sq2.log 2>&1
python - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/sq1", "gpurun_out/sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    for k, v in acc.items():
        if "ojphgpu" in k or "anonymous" in k:
            n = len(cnt[k])
            print(k, " launches", n, " ".join("%s=%.3g" % (c, x / n) for c, x in sorted(v.items())))
PY
find gpurun_out/sq1 gpurun_out/sq2 -type f -size +4M -delete
