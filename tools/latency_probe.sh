#!/bin/bash
# one block per CU or so (12 tiles): a workgroup's life against its targets per block
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for c in ${1:-il eq}; do for tpb in 8 16 32 64; do for t in 6 12; do
  python3 tools/mode_probe.py --case $c --tiles $t --option targets_per_block=$tpb | python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('%-6s tpb %2d tiles %3d  %.4f ms' % (d['case'], $tpb, d['tiles'], d['kernel_ms']))
"
done; done; done 2>&1 | tee -a gpurun_out/latency_probe.log
