#!/usr/bin/env python3
"""Levenshtein <= 2 (the reference's default) on the bench workload: the closed-form queue kernel
(lev2_stream.inc) against the banded-DP one, first-round lengths, both layouts; counters compared."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from well_duplicates_amd import synth, workload  # noqa: E402
from well_duplicates_amd.scanner import Scanner, TileBatch  # noqa: E402

sc = Scanner(0)
rows, cols = workload.HISEQ4000_ROWS, workload.HISEQ4000_COLS
centre, lvl_off, nbr = workload.honeycomb_targets(rows, cols, 2500, 5, seed=13)
sc.set_targets(centre, lvl_off, nbr)
n = rows * cols
tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 96
spec = synth.SynthSpec(seed=2, n_clusters=n, row=cols)
ids = [(1, int(t)) for t in workload.tiles_for_stype("hiseq_x")[:tiles]]
ref = None
for stride in (1, 4):
    tb = TileBatch(sc, tiles, 50, n, interleave=stride)
    tb.fill_synthetic(spec, ids, list(range(50)))
    sc.set_option("well_stride", stride)
    for closed in (0, 1):
        sc.set_option("lev2_closed", closed)
        for first in ((0,) if (stride == 4 or not closed) else (4, 5, 6)):
            sc.set_option("queue_first", first)
            blocks, _ = tb.count(2, 2)
            if ref is None:
                ref = blocks
            same = bool((blocks == ref).all())
            sc.set_option("profile", 1)
            sc.profile_reset()
            for _ in range(10):
                tb.count(2, 2)
            ms, cnt = sc.profile_get()
            sc.set_option("profile", 0)
            print("layout %s  %s  first %d: %.4f ms  same counters %s" % (
                "planes" if stride == 1 else "interleaved", "closed form" if closed else "banded DP  ", first, ms / cnt, same),
                flush=True)
    sc.set_option("queue_first", 0)
    sc.set_option("well_stride", 1)
    tb.free()
sc.close()
