#!/bin/bash
# counter passes for the sparse (sampled-targets) kernels of the bench workload
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
for c in ${CASES:-eq ham2 lev2 il}; do
  python3 tools/pmc_collect.py --case $c --tag ${TAG:-r02} --out gpurun_out/pmc > gpurun_out/pmc_$c.log 2>&1
  tail -n 12 gpurun_out/pmc_$c.log
done
