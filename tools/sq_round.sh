#!/bin/bash
# SQ counters (instruction mix + stall composition), one pass of 8 SQ counters, --pmc + --kernel-trace only.
#   tools/sq_round.sh [workload [container bits]]    (default: the 8K bench frame in 16-bit containers)
# The result is MERGED into gpurun_out/sq_counters.json under the workload's name (same kernel digest only).
set -u
WL=${1:-c3_8k_444_12b_irv97}; CT=${2:-16}; export WL
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/sq1 gpurun_out/sq2
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT --kernel-trace --output-format csv -d gpurun_out/sq1 -o sq -- env OJPH_BENCH_NOCHECK=1 python bench.py --workload $WL --container $CT --steps 2 --warmup 1 --no-cpu-baseline --plain > gpurun_out/sq1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/sq2 -o sq -- env OJPH_BENCH_NOCHECK=1 python bench.py --workload $WL --container $CT --steps 2 --warmup 1 --no-cpu-baseline --plain > gpurun_out/sq2.log 2>&1
python - <<'PY'
import csv, glob, collections, json
out = {}
for d in ("gpurun_out/sq1", "gpurun_out/sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            per[k][r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, disp in per.items():
        if "ojphgpu" not in k and "anonymous" not in k:
            continue
        n = len(disp)
        tot = collections.defaultdict(float)
        for dd in disp.values():
            for c, x in dd.items(): tot[c] += x
        print(k[:70], " launches", n, " ".join("%s=%.4g" % (c, x / n) for c, x in sorted(tot.items())))
        if d.endswith("sq1") and ("ht_encode_kernel" in k or "ht_dec_step2" in k):
            # per frame these stages are two launches: the lower resolutions' blocks (fewer wavefronts) and the top
            # resolution's; the bench's one-launch-over-all-blocks objects (most wavefronts) are left out
            groups = collections.defaultdict(list)
            for dd in disp.values(): groups[int(dd.get("SQ_WAVES", 0))].append(dd)
            sizes = sorted(groups)
            if len(sizes) >= 3: sizes = sizes[:2]
            avg = lambda g, c: sum(dd.get(c, 0.0) for dd in groups[g]) / len(groups[g])
            if "ht_encode_kernel" in k and len(sizes) == 2:
                out["ht_encode[lower resolutions]"] = {"valu_insts": avg(sizes[0], "SQ_INSTS_VALU"), "salu_insts": avg(sizes[0], "SQ_INSTS_SALU"), "wavefronts": sizes[0]}
                out["ht_encode[top resolution, side stream]"] = {"valu_insts": avg(sizes[1], "SQ_INSTS_VALU"), "salu_insts": avg(sizes[1], "SQ_INSTS_SALU"), "wavefronts": sizes[1]}
            if "ht_encode_kernel" in k and len(sizes) == 1:   # (frame batches: one launch over all blocks)
                out["ht_encode"] = {"valu_insts": avg(sizes[0], "SQ_INSTS_VALU"), "salu_insts": avg(sizes[0], "SQ_INSTS_SALU"), "wavefronts": sizes[0]}
            if "ht_dec_step2" in k:
                out["ht_dec_step2"] = {"valu_insts": sum(avg(g, "SQ_INSTS_VALU") for g in sizes), "salu_insts": sum(avg(g, "SQ_INSTS_SALU") for g in sizes), "wavefronts": sum(sizes)}
        for key, sub in (("ht_dec_step1", "ht_dec_step1"), ("ht_dec_prep", "ht_dec_prep"), ("ht_dec_fused(step 1 + step 2)", "ht_dec_fused_kernel")):
            if d.endswith("sq1") and sub in k:
                out[key] = {"valu_insts": tot["SQ_INSTS_VALU"] / n, "salu_insts": tot["SQ_INSTS_SALU"] / n}
        # second pass: what share of its wavefront-cycles a kernel spends parked at s_waitcnt (SQ_WAIT_ANY / SQ_WAVE_CYCLES, over
        # all its launches): above one half the launch is latency-bound whatever its instruction count says
        if d.endswith("sq2") and tot.get("SQ_WAVE_CYCLES"):
            share = {"wait_share": round(tot["SQ_WAIT_ANY"] / tot["SQ_WAVE_CYCLES"], 4), "issue_stall_share": round(tot["SQ_WAIT_INST_ANY"] / tot["SQ_WAVE_CYCLES"], 4)}
            for key, sub in (("ht_encode", "ht_encode_kernel"), ("ht_dec_fused(step 1 + step 2)", "ht_dec_fused_kernel"), ("ht_dec_step1", "ht_dec_step1"),
                             ("ht_dec_step2", "ht_dec_step2")):
                if sub in k:
                    out.setdefault("_waits", {})[key] = share
import os, sys
sys.path.insert(0, os.getcwd())
from openjph_amd.build import kernel_sources_digest
dig = kernel_sources_digest()
try:
    allw = json.load(open("gpurun_out/sq_counters.json"))
    if allw.get("_kernels_sha256") != dig: allw = {}
except Exception:
    allw = {}
allw["_kernels_sha256"] = dig
allw[os.environ["WL"]] = dict(out, _note="wavefront instructions per launch (per frame for multi-launch stages), SQ_INSTS_VALU / SQ_INSTS_SALU summed over the dispatch, rocprofv3 --pmc pass of tools/sq_round.sh")
json.dump(allw, open("gpurun_out/sq_counters.json", "w"), indent=1)
PY
find gpurun_out/sq1 gpurun_out/sq2 -type f -size +4M -delete
