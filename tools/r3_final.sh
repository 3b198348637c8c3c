#!/bin/bash
# round 3: the whole GPU suite on the final build (per-test timeouts)
mkdir -p gpurun_out/r3f
timeout 900 python -m pytest tests -m gpu -x -q --timeout 300 > gpurun_out/r3f/pytest_gpu_full.txt 2>&1
grep -E "passed|failed|error" gpurun_out/r3f/pytest_gpu_full.txt | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -E "smoke ok|Error|error" | tail -2
