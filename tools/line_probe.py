#!/usr/bin/env python3
"""The line walk (option line_walk: k_scan_lines, the pairs in the order of their neighbour wells) against
the queue kernel (k_scan_q, target by target): same tally blocks, per-target counts and hit records, and the
kernel times.  Usage: line_probe.py [bench|novaseq] [tiles=96]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from well_duplicates_amd import synth, workload  # noqa: E402
from well_duplicates_amd.scanner import Scanner, TileBatch  # noqa: E402

NOVA = len(sys.argv) > 1 and sys.argv[1] == "novaseq"
tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 96
sc = Scanner(0)
sc.set_option("line_pairs", int(os.environ.get("WD_LINE_PAIRS", "0")))
if NOVA:
    from well_duplicates_amd import cluster_indexes
    rows, cols = workload.NOVASEQ_ROWS, workload.NOVASEQ_COLS
    x, y = synth.honeycomb_pixels(rows, cols)
    sc.targets_from_coords(x, y, cluster_indexes.sample_centres(rows * cols, 10000, 13), levels=7,
                           max_dists=cluster_indexes.max_dists_for(7))
else:
    rows, cols = workload.HISEQ4000_ROWS, workload.HISEQ4000_COLS
    n_targets, n_levels = int(os.environ.get("WD_LINE_TARGETS", "2500")), int(os.environ.get("WD_LINE_LEVELS", "5"))
    if n_levels <= 5:
        centre, lvl_off, nbr = workload.honeycomb_targets(rows, cols, n_targets, n_levels, seed=13)
        sc.set_targets(centre, lvl_off, nbr)
    else:
        from well_duplicates_amd import cluster_indexes
        x, y = synth.honeycomb_pixels(rows, cols)
        sc.targets_from_coords(x, y, cluster_indexes.sample_centres(rows * cols, n_targets, 13), levels=n_levels,
                               max_dists=cluster_indexes.max_dists_for(n_levels))
    _c, _o, _n = sc.get_targets()
    T, levels, P = _c.shape[0], _o.shape[1] - 1, _n.shape[0]
    print("%d targets x %d levels: %d pairs, %.3f per well" % (T, levels, P, P / (rows * cols)), flush=True)
n = rows * cols
tb = TileBatch(sc, tiles, 50, n)
tb.fill_synthetic(synth.SynthSpec(seed=2 if not NOVA else 4, n_clusters=n, row=cols, plant_per_64k=1311),
                  [(1, int(t)) for t in workload.tiles_for_stype(workload.NOVASEQ_STYPE if NOVA else "hiseq_x")[:tiles]],
                  list(range(50)))
cap = 4000000
import time  # noqa: E402
sc.set_option("line_walk", 1)
t0 = time.perf_counter()
tb.count(0, 0)
t1 = time.perf_counter()
tb.count(0, 0)
t2 = time.perf_counter()
print("line tables: %.1f ms to build (first scan %.1f ms, second %.1f ms), %d blocks"
      % ((t1 - t0 - (t2 - t1)) * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3, sc.get_option("line_walk_blocks")), flush=True)
MODES = ((0, 0, "equality"), (1, 1, "hamming<=1"), (1, 2, "hamming<=2"), (2, 2, "lev<=2"))
if os.environ.get("WD_LINE_MODES"):
    MODES = [m for m in MODES if m[2] in os.environ["WD_LINE_MODES"].split(",")]
for mode, k, name in MODES:
    ref = None
    for lw in (0, 1, 1):
        sc.set_option("line_walk", lw)
        sc.hitlog_enable(cap)
        blocks, pt = tb.count(mode, k, per_target=True)
        hits, total = sc.hitlog_fetch(cap)
        sc.hitlog_enable(0)
        hl = np.sort(hits, order=["tile", "target", "slot"])
        if ref is None:
            ref = (blocks, pt, hl, total)
        same = bool((blocks == ref[0]).all() and (pt == ref[1]).all() and total == ref[3] and (hl == ref[2]).all())
        sc.set_option("profile", 1)
        sc.profile_reset()
        for _ in range(10):
            tb.count(mode, k)
        ms, cnt = sc.profile_get()
        sc.set_option("profile", 0)
        print("%-12s line_walk %d: %.4f ms  same counters, per-target counts and %d hit records %s  [%s] blocks %d"
              % (name, lw, ms / max(1, cnt), total, same, sc.last_kernel(), sc.get_option("line_walk_blocks")), flush=True)
sc.set_option("line_walk", 0)
sc.close()
