#!/bin/bash
# The frame pipelines over the workloads, with and without this round's two changes (tools/e2e_pipeline.py per line).
for cfg in "2 0" "1 0" "2 1"; do
  set -- $cfg
  export OJPHGPU_DEC_PIPE_OBJECTS=$1
  if [ $2 = 1 ]; then export OJPHGPU_COPY_PRIO_OFF=1; else unset OJPHGPU_COPY_PRIO_OFF; fi
  for w in c2:8 c5:16 c3:16; do
    timeout 300 python tools/e2e_pipeline.py --workload ${w%%:*} --container ${w##*:} --frames 100 --depth 6 --threads 4 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); e=d['encode']; c=d['decode']
print('decoder objects $1, copy-out priority %s | %-28s | encode %7.0f Msamples/s %.3f ms/frame (Tier-2 %.2f ms) | decode %7.0f Msamples/s %.3f ms/frame (parse %.2f ms) | PCIe %s' % ('off' if '$2' == '1' else 'high', d['workload'], e['Msamples_s'], e['ms_per_frame'], e['host_tier2_ms'], c['Msamples_s'], c['ms_per_frame'], c['host_parse_ms'], d['pcie_GBps']))"
  done
done
