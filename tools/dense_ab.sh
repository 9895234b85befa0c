#!/bin/bash
# A/B of dense-unit variants (tools/build_variant.py <tag> dense ...) by per-kernel times: bash tools/dense_ab.sh "<tags>"
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out
for t in base $1; do
  if [ $t = base ]; then unset WELLDUP_LIB; else export WELLDUP_LIB=$R/well_duplicates_amd/build_variants/libwelldup_$t.so; fi
  echo "#### $t"; bash $R/tools/dense_stats.sh 1 8
done 2>&1 | tee -a $R/gpurun_out/dense_ab.log
