#!/bin/bash
# Per-kernel times of the dense chain (rocprofv3 --kernel-trace --stats), equality and Levenshtein <= 2:
#   bash tools/dense_stats.sh [sym=1] [tiles=8]      (MODES="0:0" for equality only; WELLDUP_LIB=<variant> for an A/B)
cd /tmp && export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; SYM=${1:-1}; TILES=${2:-8}
for cfg in ${MODES:-0:0 2:2}; do set -- ${cfg/:/ }
rm -rf /tmp/prof_$1; rocprofv3 --kernel-trace --stats -d /tmp/prof_$1 -o run --output-format csv -- python3 $R/tools/dense_probe.py --tiles $TILES --mode $1 -k $2 --chunk 8 --sym $SYM --plant ${PLANT:-1311} > /tmp/probe_$1.log 2>&1
f=$(find /tmp/prof_$1 -name "*kernel_stats.csv" | head -1); echo "== mode $1 sym $SYM"; grep "^stream" /tmp/probe_$1.log | tail -1; grep "^scan" /tmp/probe_$1.log | tail -1
python3 - "$f" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(anonymous namespace\)::|void |\(.*\)$", "", r["Name"])
    if "dense" in n and "windows" not in n and "symcheck" not in n:
        print("  %-36s calls %s avg %8.1f us" % (n, r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
