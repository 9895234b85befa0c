#!/bin/bash
# A/B of library variants (tools/build_variant.py) on the sampled cases: bash tools/ab.sh "<tags>" "<cases>" [workload]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
tags=$1; cases=${2:-"eq lev2 il il_lev2"}; wl=${3:-sparse}
for c in $cases; do for t in base $tags; do
  if [ $t = base ]; then unset WELLDUP_LIB; else export WELLDUP_LIB=$PWD/well_duplicates_amd/build_variants/libwelldup_$t.so; fi
  python3 tools/mode_probe.py --case $c --workload $wl ${PROBE_OPTS} | python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('%-8s %-16s %-8s %.4f ms  alg/peak %.3f  %s' % ('$wl', d['case'], '$t', d['kernel_ms'], d['alg_bytes_over_peak'], d['kernel']))
"
done; done 2>&1 | tee -a gpurun_out/ab.log
