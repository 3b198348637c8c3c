#!/bin/bash
# Where does the block encoder's time go at a given quantisation step?  Alternates library builds made with the encoder's
# ablation switches (tools/build_variant.py <name> kernels_ht_enc.hip -DABL=<bits>; results of those builds are WRONG by
# design, only their times count) over encode-only runs of the 8K frame:   tools/enc_qstep_ab.sh "orig abl1 ..." 0.01 0.001
VARS=$1; shift
O=gpurun_out/enc_qstep_ab.txt; : > $O
cp openjph_amd/libojphgpu.so /tmp/lib_orig.so
for q in "$@"; do
  for rep in 1 2; do
    for v in $VARS; do
      if [ $v = orig ]; then cp /tmp/lib_orig.so openjph_amd/libojphgpu.so; else cp openjph_amd/variants/lib_$v.so openjph_amd/libojphgpu.so; fi
      timeout 120 python - $q $v >> $O 2>/tmp/err.txt <<'PY' || tail -2 /tmp/err.txt >> $O
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from bench import WORKLOADS, workload_image
from openjph_amd import codec
from openjph_amd.plan import Plan, make_params
q, v = float(sys.argv[1]), sys.argv[2]
name = "c3_8k_444_12b_irv97"
w, h, nc, bd, rev, ct, _, tile = WORKLOADS[name]
d = torch.from_numpy(workload_image(name).astype(np.int16)).cuda()
enc = codec.Encoder(plan=Plan(make_params(w, h, nc, bit_depth=bd, reversible=rev, qstep=q)))
for _ in range(30):
    enc.run_device(d)
torch.cuda.synchronize()
ht = np.zeros(2); tot = 0.0
for _ in range(20):
    enc.run_device(d); t = enc.timing(); ht += np.array(t["ht_launches_ms"][:2]); tot += t["total_ms"]
print("qstep %.3f %-6s encode %.3f ms  ht launches %.3f %.3f" % (q, v, tot / 20, ht[0] / 20, ht[1] / 20), flush=True)
PY
    done
  done
done
cp /tmp/lib_orig.so openjph_amd/libojphgpu.so
cat $O
