#!/bin/bash
# gpurun_out/pmc/<TAG>_* (tools/pmc_all.sh) -> profiles/<TAG>_{pmc,kernel_stats}_*, the resource table, traffic.json:
#   bash tools/adopt_evidence.sh r04_r [tag of the files this pass supersedes, e.g. r04_p]
set -e
cd "$(dirname "$0")/.."
T=$1; OLD=$2; S=gpurun_out/pmc
for c in eq ham2 lev2 il il_lev2; do cp $S/${T}_$c.json profiles/${T}_pmc_sparse_$c.json; cp $S/${T}_${c}_kt_run_kernel_stats.csv profiles/${T}_kernel_stats_sparse_$c.csv; done
for c in eq lev2 il il_lev2; do cp $S/${T}_${c}_workloadnovaseq.json profiles/${T}_pmc_novaseq_$c.json; cp $S/${T}_${c}_workloadnovaseq_kt_run_kernel_stats.csv profiles/${T}_kernel_stats_novaseq_$c.csv; done
for c in eq ham2 lev2; do cp $S/${T}_dense_${c}_tiles8.json profiles/${T}_pmc_dense_${c}_8tiles_2pct.json; cp $S/${T}_dense_${c}_tiles8_kt_run_kernel_stats.csv profiles/${T}_kernel_stats_dense_${c}_8tiles_2pct.csv; done
cp $S/${T}_kernel_resources.json profiles/${T%_*}_kernel_resources.json
cp $S/${T}_kernel_resources.txt profiles/${T%_*}_kernel_resources.txt
if [ -n "$OLD" ]; then git rm -q --ignore-unmatch profiles/${OLD}_pmc_* profiles/${OLD}_kernel_stats_sparse_* profiles/${OLD}_kernel_stats_novaseq_* profiles/${OLD}_kernel_stats_dense_*; fi
python3 tools/make_traffic.py | tail -3
python3 - <<'PY'
import json
from well_duplicates_amd import _lib
ids = _lib.build_ids()
t = json.load(open("profiles/traffic.json"))
stale = [k for k, v in t.items() if isinstance(v, dict) and v.get("unit_id") and v["unit_id"] != ids.get(v.get("unit"))]
print(len(t), "entries of traffic.json; of another build than this tree's library:", stale)
PY
