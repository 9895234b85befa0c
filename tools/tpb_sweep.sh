#!/bin/bash
# equality on the bench workload against targets per workgroup, at a lane and at four lanes per launch: bash tools/tpb_sweep.sh
cd $GRAFT_REPO_ROOT
for tiles in 96 384; do for tpb in 16 32 48 64; do
python3 tools/mode_probe.py --case eq --tiles $tiles --option targets_per_block=$tpb --reps 8 | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('tiles $tiles tpb $tpb: %.4f ms  %.3f' % (d['kernel_ms'], d['alg_bytes_over_peak']))
"
done; done
