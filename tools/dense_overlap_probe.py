#!/usr/bin/env python3
"""The dense path (every well a centre, 3 levels, 150 bp, 2 % planted) with and without the part
pipeline over two streams (option dense_overlap), part sizes, the three compare modes: ms per tile,
and the counter blocks compared with the one-stream run."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from well_duplicates_amd import synth, workload  # noqa: E402
from well_duplicates_amd.scanner import Scanner, TileBatch  # noqa: E402

tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 16
parts = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
rows, cols, levels, L = workload.HISEQ4000_ROWS, workload.HISEQ4000_COLS, 3, 150
n = rows * cols
x, y = synth.honeycomb_pixels(rows, cols)
sc = Scanner(0)
T, P = sc.targets_from_coords(x, y, None, levels=levels)
spec = synth.SynthSpec(seed=5, n_clusters=n, row=cols, plant_per_64k=1311)
tb = TileBatch(sc, tiles, L, n)
tb.fill_synthetic(spec, [(1 + i // 96, 1101 + i % 96) for i in range(tiles)], list(range(L)))
ncnt = 1 + 5 * levels
out = sc.malloc(tiles * ncnt * 8)
MODES = ((0, 0, "equality"), (1, 2, "hamming<=2"), (2, 2, "levenshtein<=2"))
if len(sys.argv) > 3:
    MODES = [m for m in MODES if m[2].startswith(sys.argv[3])]
NO_BASE = os.environ.get("WD_PROBE_NO_BASELINE") == "1"
PACK_BLOCKS = [int(v) for v in os.environ.get("WD_PROBE_PACK_BLOCKS", "1024").split(",")]
for mode, k, name in MODES:
    ref = None
    for overlap, part, pb in ([] if NO_BASE else [(0, 0, 0)]) + [(1, p, b) for p in parts for b in PACK_BLOCKS]:
        sc.set_option("dense_overlap", overlap)
        sc.set_option("dense_pack_blocks", pb)
        sc.set_option("dense_part_tiles", part)
        sc.scan_async(tb.tables, tiles, L, n, mode, k, out)
        sc.set_option("profile", 1)
        sc.profile_reset()
        for _ in range(4):
            sc.scan_async(tb.tables, tiles, L, n, mode, k, out)
        ms, cnt = sc.profile_get()
        sc.set_option("profile", 0)
        blk = sc.d2h(out, tiles * ncnt * 8, np.int64)
        if ref is None:
            ref = blk
        print("%-15s overlap %d part %2d pack blocks %5d: %.4f ms per tile   same counters %s" % (
            name, overlap, part, pb, ms / cnt / tiles, bool((blk == ref).all())), flush=True)
sc.close()
