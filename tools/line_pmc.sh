# Counters of the line walk (k_scan_lines) and the queue kernel on one workload: bash tools/line_pmc.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "FETCH_SIZE TCC_EA0_RDREQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  d=/tmp/lwp_$(echo $pass | cut -c1-6 | tr -d ' '); rm -rf $d
  WD_LINE_PAIRS=12288 rocprofv3 --pmc $pass --kernel-trace -d $d -o run --output-format csv -- python3 $R/tools/line_probe.py ${WD_LINE_WORKLOAD:-novaseq} 96 > $d.log 2>&1
  if [ -z "$(find $d -name '*counter_collection.csv' 2>/dev/null | head -1)" ]; then echo "no counters for pass [$pass]: see $d.log"; tail -5 $d.log; exit 1; fi
  python3 - $d <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_scan_lines" in n or "k_scan_q" in n:
            import re
            per[(re.search(r"k_scan_\w+<[^>]*>", n).group(0), r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (n, d, c), v in per.items():
        acc[n][c].append(v)
for n, cs in sorted(acc.items()):
    print(n, {c: round(sum(v[1:]) / max(1, len(v) - 1)) for c, v in cs.items()})
PY
done
