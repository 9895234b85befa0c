#!/usr/bin/env python3
"""The lazy scan's worst case on real low-diversity input (amplicon-like: most neighbours equal
their centre, nothing dies early): kernel time per tile in the plane layout and in the interleaved
resident layout, equality and Levenshtein <= 2, against ordinary reads."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from well_duplicates_amd import synth, workload  # noqa: E402
from well_duplicates_amd.scanner import Scanner, TileBatch  # noqa: E402

tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rows, cols = workload.HISEQ4000_ROWS, workload.HISEQ4000_COLS
n = rows * cols
centre, lvl_off, nbr = workload.honeycomb_targets(rows, cols, 2500, 5, seed=13)
L = 50
sc = Scanner(0)
sc.set_option("line_walk", int(os.environ.get("WD_LINE_WALK", "-1")))      # 1: the pairs in the order of their neighbour wells
sc.set_targets(centre, lvl_off, nbr)
for label, kw in (("ordinary reads", dict()), ("every read equal (all no-calls)", dict(nocall_per_64k=65536))):
    spec = synth.SynthSpec(seed=2, n_clusters=n, row=cols, **kw)
    for interleave in (1, 4):
        tb = TileBatch(sc, tiles, L, n, interleave=interleave)
        tb.fill_synthetic(spec, [(1, 1101 + i) for i in range(tiles)], list(range(L)))
        sc.set_option("well_stride", interleave)
        for mode, k, name in ((0, 0, "equality"), (2, 2, "Levenshtein<=2")):
            blocks, _ = tb.count(mode, k)
            sc.set_option("profile", 1)
            sc.profile_reset()
            for _ in range(5):
                blocks, _ = tb.count(mode, k)
            ms, cnt = sc.profile_get()
            sc.set_option("profile", 0)
            dups = int(np.asarray(blocks)[:, 6:11].sum())
            wells = int(np.asarray(blocks)[:, 1:6].sum())
            print("%-52s %-12s %-15s %8.3f ms per %d tiles = %7.2f us/tile, dups/compares %.3f  [%s]"
                  % (label, "interleaved" if interleave == 4 else "planes", name, ms / cnt, tiles, ms / cnt / tiles * 1e3,
                     dups / max(1, wells), sc.last_kernel()), flush=True)
        sc.set_option("well_stride", 1)
        tb.free()
