#!/bin/bash
# wavefront-instruction counts of the block encoder per library variant (one launch over all blocks of the bench frame):
#   tools/enc_counters.sh orig abl1 ...      -> VALU / SALU / LDS instructions per code-block
set -u
export TMPDIR=/tmp
cp openjph_amd/libojphgpu.so /tmp/lib_cnt_orig.so
for v in "$@"; do
  if [ $v = orig ]; then cp /tmp/lib_cnt_orig.so openjph_amd/libojphgpu.so; else cp openjph_amd/variants/lib_$v.so openjph_amd/libojphgpu.so; fi
  rm -rf /tmp/cnt_$v
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d /tmp/cnt_$v -o c -- python tools/enc_run.py 2 > /tmp/cnt_$v.log 2>&1
  python - $v <<'PY'
import csv, glob, sys, collections
v = sys.argv[1]
per = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("/tmp/cnt_%s/**/*counter_collection.csv" % v, recursive=True):
    for r in csv.DictReader(open(f)):
        if "ht_encode_kernel" in r["Kernel_Name"]:
            per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
if not per:
    print(v, "no counters"); sys.exit(0)
d = list(per.values())[-1]
w = d["SQ_WAVES"]
print("%-8s per block: VALU %.0f  SALU %.0f  LDS %.0f   (%d wavefronts)" % (v, d["SQ_INSTS_VALU"] / w, d["SQ_INSTS_SALU"] / w, d["SQ_INSTS_LDS"] / w, w))
PY
done
cp /tmp/lib_cnt_orig.so openjph_amd/libojphgpu.so
