#!/bin/bash
# end-to-end (file to file) timings of the CLI tools on the 8K C3 frame, next to the reference CLI figures of BASELINE.md
python - <<'PY'
import numpy as np, sys, os
sys.path.insert(0, os.getcwd())
from tests.synth import synth_image
img = synth_image(3, 4320, 7680, 12, seed=1234)
img.astype("<u2").tofile("/tmp/c3.yuv")
PY
for i in 1 2 3; do OJPHGPU_TIMING=1 ./openjph_amd/apps/ojph_compress -i /tmp/c3.yuv -o /tmp/c3.j2c -qstep 0.001 -dims "{7680,4320}" -num_comps 3 -signed false -bit_depth 12 -downsamp "{1,1}" 2>&1 | grep -v amdgpu.ids; done
for i in 1 2; do ./openjph_amd/apps/ojph_expand -i /tmp/c3.j2c -o /tmp/c3_back.yuv 2>&1 | grep -v amdgpu.ids; done
ls -la /tmp/c3.j2c
