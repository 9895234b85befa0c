#!/usr/bin/env python3
"""Probe of the dense configuration (BASELINE config 5): every well of a full-size tile is a
centre.  Times the device neighbour generator and the scan on a few tiles."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from well_duplicates_amd import synth, workload
from well_duplicates_amd.scanner import Scanner, TileBatch

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=workload.HISEQ4000_ROWS)
ap.add_argument("--cols", type=int, default=workload.HISEQ4000_COLS)
ap.add_argument("--levels", type=int, default=3)
ap.add_argument("--bases", type=int, default=150)
ap.add_argument("--tiles", type=int, default=2)
ap.add_argument("--mode", type=int, default=0)
ap.add_argument("-k", type=int, default=0)
ap.add_argument("--chunk", default="4", help="comma list of dense_tile_chunk values to time")
ap.add_argument("--pack", type=int, default=-1, help="dense_pack option: -1 auto, 0 never, 1 always")
ap.add_argument("--sym", type=int, default=1, help="dense_sym option: 1 = pairs compared from their lower well only")
ap.add_argument("--plant", type=int, default=1311, help="planted wells per 65536 (1311 = 2 %%)")
a = ap.parse_args()
n = a.rows * a.cols
x, y = synth.honeycomb_pixels(a.rows, a.cols)
sc = Scanner(0)
sc.set_option("dense_sym", a.sym)
t0 = time.time()
T, P = sc.targets_from_coords(x, y, None, levels=a.levels)
print("generator: %d centres, %d slots (%.1f per centre) in %.2f s" % (T, P, P / T, time.time() - t0))
spec = synth.SynthSpec(seed=5, n_clusters=n, row=a.cols, plant_per_64k=a.plant)
tb = TileBatch(sc, a.tiles, a.bases, n)
tb.fill_synthetic(spec, [(1, 1101 + i) for i in range(a.tiles)], list(range(a.bases)))
out = sc.malloc(a.tiles * (1 + 5 * a.levels) * 8)
print("stream read of the resident planes: %.0f GB/s" % sc.stream_read_gbs(tb.d_planes, tb.plane_bytes - tb.plane_bytes % 16, 3))
sc.set_option("profile", 1)
sc.set_option("dense_pack", a.pack)
for ch in [int(v) for v in a.chunk.split(",")] * 2:
    sc.set_option("dense_tile_chunk", ch)
    print("dense_tile_chunk", ch, end=": ")
    sc.scan_async(tb.tables, a.tiles, a.bases, n, a.mode, a.k, out)
    sc.profile_reset()
    for _ in range(4):
        sc.scan_async(tb.tables, a.tiles, a.bases, n, a.mode, a.k, out)
    ms, cnt = sc.profile_get()
    ms /= max(1, cnt)
    blk = sc.d2h(out, a.tiles * (1 + 5 * a.levels) * 8, np.int64).reshape(a.tiles, -1)
    C = int(blk[:, 1:1 + a.levels].sum()); Tv = int(blk[:, 0].sum())
    b_dense = a.tiles * (n * a.bases + 4 * n * (1 + P / T) + n)
    print(sc.last_kernel())
    print("scan: %.3f ms for %d tiles, %d compares -> %.1f Gcmp/s; dense B_alg %.2f GB -> %.0f GB/s"
          % (ms, a.tiles, C, C / ms / 1e6, b_dense / 1e9, b_dense / ms / 1e6))
