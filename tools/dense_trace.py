#!/usr/bin/env python3
"""Timeline of one dense scan from a rocprofv3 --kernel-trace CSV: which kernels ran when, on which
queue, and how much of the scan two kernels were running side by side.

    rocprofv3 --kernel-trace -d DIR -o run --output-format csv -- python3 tools/dense_overlap_probe.py 16 0 eq
    python3 tools/dense_trace.py DIR [last-N-scans]
"""
import csv
import glob
import os
import re
import sys


def main():
    d = sys.argv[1]
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
                name = re.sub(r"^void ", "", name)
                name = re.sub(r"\(.*\)$", "", name)
                if name.startswith("k_dense"):
                    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?")))
    rows.sort()
    # the last scan: from the last k_dense_sig that follows a k_dense_reduce gap
    reduces = [i for i, r in enumerate(rows) if r[2].startswith("k_dense_reduce")]
    n_red = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    first = reduces[-n_red - 1] + 1 if len(reduces) > n_red else 0
    sel = rows[first:]
    t0 = sel[0][0]
    busy1 = busy2 = 0
    events = sorted([(s, 1) for s, e, _, _ in sel] + [(e, -1) for s, e, _, _ in sel])
    depth, last = 0, t0
    for t, dlt in events:
        if depth == 1:
            busy1 += t - last
        elif depth >= 2:
            busy2 += t - last
        depth += dlt
        last = t
    for s, e, name, q in sel:
        print("%9.1f us  +%8.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, name))
    span = sel[-1][1] - t0
    print("span %.1f us: one kernel running %.1f us, two or more %.1f us, none %.1f us" % (
        span / 1e3, busy1 / 1e3, busy2 / 1e3, (span - busy1 - busy2) / 1e3))


if __name__ == "__main__":
    main()
