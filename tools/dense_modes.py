#!/usr/bin/env python3
"""Does the dense chain's time depend on where its buffers lie?  One process, several trials: a dummy
allocation of a different size in front, the tiles allocated and filled anew, the same scan timed.
(k_dense_pack runs 10 % apart from process to process on some boxes: profiles/r04_o_*.)

    python tools/dense_modes.py [trials=6] [tiles=8]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from well_duplicates_amd import synth, workload                      # noqa: E402
from well_duplicates_amd.scanner import Scanner, TileBatch           # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 6
tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows, cols, levels, L = workload.HISEQ4000_ROWS, workload.HISEQ4000_COLS, 3, 150
n = rows * cols
x, y = synth.honeycomb_pixels(rows, cols)
sc = Scanner(0)
T, P = sc.targets_from_coords(x, y, None, levels=levels)
spec = synth.SynthSpec(seed=5, n_clusters=n, row=cols, plant_per_64k=1311)
sc.set_option("profile", 1)
first = None
for trial in range(trials):
    pad = sc.malloc(1 + trial * 777 * (1 << 20)) if trial else 0          # shifts whatever comes after it
    tb = TileBatch(sc, tiles, L, n)
    tb.fill_synthetic(spec, [(1, 1101 + i) for i in range(tiles)], list(range(L)))
    out = sc.malloc(tiles * (1 + 5 * levels) * 8)
    sc.scan_async(tb.tables, tiles, L, n, 0, 0, out)
    sc.profile_reset()
    for _ in range(5):
        sc.scan_async(tb.tables, tiles, L, n, 0, 0, out)
    ms, cnt = sc.profile_get()
    blk = sc.d2h(out, tiles * (1 + 5 * levels) * 8, np.int64)
    first = blk if first is None else first
    assert (blk == first).all()
    gbs = sc.stream_read_gbs(tb.d_planes, tb.plane_bytes - tb.plane_bytes % 16, 3)
    print("trial %d: planes at 0x%x (pad %4d MB in front)  scan %.3f ms per %d tiles   stream read %.0f GB/s"
          % (trial, tb.d_planes, (trial * 777), ms / max(1, cnt), tiles, gbs), flush=True)
    sc.free(out)
    tb.free()
    if pad:
        sc.free(pad)
sc.close()
