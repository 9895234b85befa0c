#!/bin/bash
# Kernel times of every sampled-scan case bench.py quotes (tools/mode_probe.py, HIP events): one JSON line each.
#   bash tools/probe_all.sh [sparse] [novaseq]      -> gpurun_out/probe_all.log
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
for what in ${@:-sparse novaseq}; do
  for c in eq ham2 lev2 il il_lev2; do
    python3 tools/mode_probe.py --case $c --workload $what | python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('%-8s %-16s %.4f ms  alg/peak %.3f  %s' % ('$what', d['case'], d['kernel_ms'], d['alg_bytes_over_peak'], d['kernel']))
"
  done
done 2>&1 | tee gpurun_out/probe_all.log
