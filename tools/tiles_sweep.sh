#!/bin/bash
# kernel time against the number of tiles of a scan: bash tools/tiles_sweep.sh "<cases>" "<tile counts>"
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for c in ${1:-il}; do for t in ${2:-12 24 48 96 192 384}; do
  python3 tools/mode_probe.py --case $c --tiles $t ${PROBE_OPTS} | python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('%-10s tiles %4d  %.4f ms  %.3f us/tile  alg/peak %.3f' % (d['case'], d['tiles'], d['kernel_ms'], d['kernel_ms']*1e3/d['tiles'], d['alg_bytes_over_peak']))
"
done; done 2>&1 | tee -a gpurun_out/tiles_sweep.log
