#!/bin/bash
# SQ wait / busy counters of the block encoder (one launch over all blocks of the bench frame), per library variant
set -u
export TMPDIR=/tmp
cp openjph_amd/libojphgpu.so /tmp/lib_w_orig.so
for v in "$@"; do
  if [ $v = orig ]; then cp /tmp/lib_w_orig.so openjph_amd/libojphgpu.so; else cp openjph_amd/variants/lib_$v.so openjph_amd/libojphgpu.so; fi
  rm -rf /tmp/w_$v
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d /tmp/w_$v -o c -- python tools/enc_run.py 2 > /tmp/w_$v.log 2>&1
  rm -rf /tmp/w2_$v
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/w2_$v -o c -- python tools/enc_run.py 2 > /tmp/w2_$v.log 2>&1
  python - $v <<'PY'
import csv, glob, sys, collections
v = sys.argv[1]
tot = {}
for d in ("/tmp/w_%s" % v, "/tmp/w2_%s" % v):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "ht_encode_kernel" in r["Kernel_Name"]:
                per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    if per:
        tot.update(list(per.values())[-1])
print(v, " ".join("%s=%.4g" % (k, x) for k, x in sorted(tot.items())))
PY
  tail -2 /tmp/w2_$v.log | cut -c1-300
done
cp /tmp/lib_w_orig.so openjph_amd/libojphgpu.so
