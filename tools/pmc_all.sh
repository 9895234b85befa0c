#!/bin/bash
# Counter passes for every kernel bench.py quotes traffic for: the sparse kernels of the bench workload,
# the BASELINE configs[3] shape and the dense chain.  Summaries (a few KB) -> gpurun_out/pmc, raw CSVs /tmp.
#   TAG=r03_c bash tools/pmc_all.sh [sparse] [novaseq] [dense]
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
RAW=/tmp/wd_pmc_raw
TAG=${TAG:-r04}
mkdir -p gpurun_out/pmc $RAW
run() {   # case, extra probe args
  python3 tools/pmc_collect.py --case $1 --tag $TAG --out $RAW ${2:+--probe-args "$2"} > gpurun_out/pmc_$1_$3.log 2>&1 || true
  tail -n 6 gpurun_out/pmc_$1_$3.log
}
for what in ${@:-sparse novaseq dense}; do
  case $what in
    sparse)  for c in eq ham2 lev2 il il_lev2; do run $c "" sparse; done ;;
    novaseq) for c in eq lev2 il il_lev2; do run $c "--workload novaseq" novaseq; done ;;
    dense)   for c in dense_eq dense_ham2 dense_lev2; do run $c "--tiles 8" dense; done ;;
  esac
done
cp $RAW/*.json gpurun_out/pmc/
find $RAW -name "*kernel_stats.csv" | while read f; do cp "$f" gpurun_out/pmc/$(echo "$f" | sed "s|$RAW/||; s|/|_|g"); done
# the resource table of the very library the counters were read from (its build ids inside)
python3 tools/kernel_resources.py --json gpurun_out/pmc/${TAG}_kernel_resources.json > gpurun_out/pmc/${TAG}_kernel_resources.txt
ls gpurun_out/pmc
