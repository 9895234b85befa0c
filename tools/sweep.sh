#!/bin/bash
# sweep one scanner option over the sampled-scan cases: bash tools/sweep.sh <option> "<values>" "<cases>" [workload]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
opt=$1; vals=$2; cases=${3:-"eq lev2 il il_lev2"}; wl=${4:-sparse}
for c in $cases; do for v in $vals; do
  python3 tools/mode_probe.py --case $c --workload $wl --option $opt=$v | python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('%-8s %-16s $opt=$v  %.4f ms  alg/peak %.3f  %s' % ('$wl', d['case'], d['kernel_ms'], d['alg_bytes_over_peak'], d['kernel']))
"
done; done 2>&1 | tee -a gpurun_out/sweep.log
