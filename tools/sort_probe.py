#!/usr/bin/env python3
"""Bench workload (96 tiles x 2500 sampled targets x 5 levels x 50 bp) with the targets walked in file
order and sorted by centre (option sort_targets): kernel time per mode, counters compared."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from well_duplicates_amd import synth, workload  # noqa: E402
from well_duplicates_amd.scanner import Scanner, TileBatch  # noqa: E402

NOVA = len(sys.argv) > 1 and sys.argv[1] == "novaseq"
sc = Scanner(0)
sc.set_option("sort_strip", int(os.environ.get("WD_SORT_STRIP", "512")))
if NOVA:
    from well_duplicates_amd import cluster_indexes
    rows, cols = workload.NOVASEQ_ROWS, workload.NOVASEQ_COLS
    x, y = synth.honeycomb_pixels(rows, cols)
    sc.targets_from_coords(x, y, cluster_indexes.sample_centres(rows * cols, 10000, 13), levels=7,
                           max_dists=cluster_indexes.max_dists_for(7))
else:
    rows, cols = workload.HISEQ4000_ROWS, workload.HISEQ4000_COLS
    centre, lvl_off, nbr = workload.honeycomb_targets(rows, cols, 2500, 5, seed=13)
    sc.set_targets(centre, lvl_off, nbr)
n = rows * cols
tiles = 96
tb = TileBatch(sc, tiles, 50, n)
tb.fill_synthetic(synth.SynthSpec(seed=2 if not NOVA else 4, n_clusters=n, row=cols),
                  [(1, int(t)) for t in workload.tiles_for_stype(workload.NOVASEQ_STYPE if NOVA else "hiseq_x")[:tiles]],
                  list(range(50)))
for mode, k, name in ((0, 0, "equality"), (1, 2, "hamming<=2"), (2, 2, "lev<=2"), (2, 3, "lev<=3")):
    ref = None
    for srt in (0, 1, 0, 1):
        sc.set_option("sort_targets", srt)
        blocks, pt = tb.count(mode, k, per_target=True)
        if ref is None:
            ref = (blocks, pt)
        same = bool((blocks == ref[0]).all() and (pt == ref[1]).all())
        sc.set_option("profile", 1)
        sc.profile_reset()
        for _ in range(10):
            tb.count(mode, k)
        ms, cnt = sc.profile_get()
        sc.set_option("profile", 0)
        print("%-12s sort_targets %d: %.4f ms  same counters and per-target counts %s  [%s]" % (
            name, srt, ms / cnt, same, sc.last_kernel()), flush=True)
sc.close()
