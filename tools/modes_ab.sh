#!/bin/bash
# tools/dense_modes.py for the shipped library and variants of it: bash tools/modes_ab.sh "<tags>" [trials]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for t in base $1; do
  if [ $t = base ]; then unset WELLDUP_LIB; else export WELLDUP_LIB=$PWD/well_duplicates_amd/build_variants/libwelldup_$t.so; fi
  echo "#### $t"; timeout -k 10 300 python3 tools/dense_modes.py ${2:-5} 8 | sed "s/planes at .* in front)//"
done
