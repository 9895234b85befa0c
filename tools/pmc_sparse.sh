#!/bin/bash
# counter passes for the sparse (sampled-targets) kernels of the bench workload; only the summaries
# (a few KB each) are kept under gpurun_out/, the raw counter CSVs go to /tmp
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
RAW=/tmp/wd_pmc_raw
mkdir -p gpurun_out/pmc $RAW
for c in ${CASES:-eq ham2 lev2 il il_lev2}; do
  python3 tools/pmc_collect.py --case $c --tag ${TAG:-r03} --out $RAW ${PROBE_ARGS:+--probe-args "$PROBE_ARGS"} > gpurun_out/pmc_$c.log 2>&1
  cp $RAW/*.json gpurun_out/pmc/
  find $RAW -name "*kernel_stats.csv" | while read f; do cp "$f" gpurun_out/pmc/$(echo "$f" | sed "s|$RAW/||; s|/|_|g"); done
  tail -n 12 gpurun_out/pmc_$c.log
done
