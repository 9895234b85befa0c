#!/usr/bin/env python3
"""Kernel time of the bench workload (96 tiles x 2500 targets x 5 levels x 50 bp) per compare mode and
first-round length of the queue kernel (option queue_first; Hamming family only)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from well_duplicates_amd import synth, workload  # noqa: E402
from well_duplicates_amd.scanner import Scanner, TileBatch  # noqa: E402

NOVA = len(sys.argv) > 1 and sys.argv[1] == "novaseq"
sc = Scanner(0)
if NOVA:                     # BASELINE configs[3] shape: 10 000 targets x 7 rings on NovaSeq tiles
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
tb.fill_synthetic(synth.SynthSpec(seed=2, n_clusters=n, row=cols),
                  [(1, int(t)) for t in workload.tiles_for_stype(workload.NOVASEQ_STYPE if NOVA else "hiseq_x")[:tiles]], list(range(50)))
MODES = ((0, 0, "equality"), (1, 1, "hamming<=1"), (1, 2, "hamming<=2"), (1, 3, "hamming<=3"), (2, 2, "lev<=2"),
         (2, 3, "lev<=3"), (2, 4, "lev<=4"), (2, 5, "lev<=5"), (2, 6, "lev<=6"), (2, 7, "lev<=7"))
if len(sys.argv) > 1 and sys.argv[1] == "ham":
    MODES = [m for m in MODES if m[0] == 1]
if len(sys.argv) > 1 and sys.argv[1] == "lev":
    MODES = [m for m in MODES if m[0] == 2]
for mode, k, name in MODES:
    for first in ((0, 1, 2, 3, 4, 5, 6, 7, 8) if mode < 2 else (0,)):
        sc.set_option("queue_first", first)
        tb.count(mode, k)
        sc.set_option("profile", 1)
        sc.profile_reset()
        for _ in range(10):
            tb.count(mode, k)
        ms, cnt = sc.profile_get()
        sc.set_option("profile", 0)
        print("%-12s queue_first %d: %.4f ms" % (name, first, ms / cnt), flush=True)
sc.set_option("queue_first", 0)
