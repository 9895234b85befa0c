#!/usr/bin/env python3
"""One compare mode of one workload, a few scans, nothing else - the program that
tools/pmc_collect.py puts under rocprofv3 (one counter pass per run), also usable alone.

Workloads
  sparse   bench.py's: 96 tiles x 2500 sampled targets x 5 levels x 50 bp (BASELINE configs[1])
  cfg4     96 tiles x 10000 targets x 7 levels x 50 bp (BASELINE configs[3] shape, HiSeq grid)
  dense    every well of a full-size tile a centre, 3 levels, 150 bp (BASELINE configs[4])

Cases: eq, ham2, lev2, full (no early exit), il (equality, cycles interleaved by four);
dense_eq, dense_ham2, dense_lev2 (with --plant wells per 65536 planted as duplicates).

Prints one JSON line: kernel ms per scan from HIP events on the launch stream, compares,
algorithmic bytes.
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from well_duplicates_amd import synth, workload                     # noqa: E402
from well_duplicates_amd.scanner import Scanner, TileBatch         # noqa: E402

SPARSE = {"eq": (0, 0), "ham2": (1, 2), "lev2": (2, 2), "lev3": (2, 3), "full": (0, 0), "il": (0, 0), "il_lev2": (2, 2)}
DENSE = {"dense_eq": (0, 0), "dense_ham2": (1, 2), "dense_lev2": (2, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", required=True)
    ap.add_argument("--workload", default="sparse", choices=["sparse", "cfg4", "novaseq"])
    ap.add_argument("--tiles", type=int, default=None)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--plant", type=int, default=1311, help="planted wells per 65536 (1311 = 2 %%)")
    ap.add_argument("--option", action="append", default=[])
    a = ap.parse_args()
    rows, cols = workload.HISEQ4000_ROWS, workload.HISEQ4000_COLS
    n = rows * cols
    sc = Scanner(0)
    if a.case in DENSE:
        mode, k = DENSE[a.case]
        levels, L = 3, 150
        tiles = a.tiles or 8
        x, y = synth.honeycomb_pixels(rows, cols)
        T, P = sc.targets_from_coords(x, y, None, levels=levels)
        spec = synth.SynthSpec(seed=5, n_clusters=n, row=cols, plant_per_64k=a.plant)
        interleave = 1
    else:
        mode, k = SPARSE[a.case]
        T, levels, L = (2500, 5, 50) if a.workload == "sparse" else (10000, 7, 50)
        tiles = a.tiles or 96
        if a.workload == "novaseq":                    # BASELINE configs[3]: NovaSeq tiles, device-generated rings
            from well_duplicates_amd import cluster_indexes
            rows, cols = workload.NOVASEQ_ROWS, workload.NOVASEQ_COLS
            n = rows * cols
            x, y = synth.honeycomb_pixels(rows, cols)
            T, P = sc.targets_from_coords(x, y, cluster_indexes.sample_centres(n, T, 13), levels=levels,
                                          max_dists=cluster_indexes.max_dists_for(levels))
            spec = synth.SynthSpec(seed=4, n_clusters=n, row=cols)
        else:
            centre, lvl_off, nbr = workload.honeycomb_targets(rows, cols, T, levels, 13)
            sc.set_targets(centre, lvl_off, nbr)
            P = int(nbr.shape[0])
            spec = synth.SynthSpec(seed=2, n_clusters=n, row=cols)
        interleave = 4 if a.case.startswith("il") else 1
    for opt in a.option:
        name, val = opt.split("=")
        sc.set_option(name, int(val))
    if a.case == "full":
        sc.set_option("early_exit", 0)
    tb = TileBatch(sc, tiles, L, n, interleave=interleave)
    tb.fill_synthetic(spec, [(1 + i // 96, 1101 + i % 96) for i in range(tiles)], list(range(L)))
    ncnt = 1 + 5 * levels
    out = sc.malloc(tiles * ncnt * 8)
    if interleave == 4:
        sc.set_option("well_stride", 4)
    sc.scan_async(tb.tables, tiles, L, n, mode, k, out)          # tables, caches, first-use allocations
    sc.scan_status()
    sc.set_option("profile", 1)
    sc.profile_reset()
    for _ in range(a.reps):
        sc.scan_async(tb.tables, tiles, L, n, mode, k, out)
    ms, cnt = sc.profile_get()
    sc.scan_status()
    blk = sc.d2h(out, tiles * ncnt * 8, np.int64).reshape(tiles, ncnt)
    C, Tv = int(blk[:, 1:1 + levels].sum()), int(blk[:, 0].sum())
    if a.case in DENSE:
        b_alg = tiles * (n * L + 4 * n * (1 + P / T) + n)
    else:
        b_alg = C * (L + 4) + Tv * (L + 5) + 8 * ncnt * tiles
    ms /= max(1, cnt)
    extra = {}
    if a.case in DENSE:
        extra = {"groups": (T + 63) // 64, "uniform_groups": sc.get_option("dense_uniform_groups"),
                 "window_groups": sc.get_option("dense_window_groups"),
                 "window_dwords": sc.get_option("dense_window_dwords")}
    case_name = ("novaseq_" + a.case) if a.workload == "novaseq" else a.case
    from well_duplicates_amd import _lib
    ids = _lib.build_ids()
    print(json.dumps({**extra, "kernel": sc.last_kernel(), "build_id": ids["all"],
                      "unit": _lib.unit_of_kernel(sc.last_kernel()), "unit_id": ids.get(_lib.unit_of_kernel(sc.last_kernel())),
                      "case": case_name, "workload": "dense" if a.case in DENSE else a.workload, "tiles": tiles,
                      "T": T, "levels": levels, "L": L, "mode": mode, "k": k, "plant_per_64k": spec.plant_per_64k,
                      "scans_timed": cnt, "kernel_ms": round(ms, 5), "compares": C, "valid_targets": Tv,
                      "duplicates": int(blk[:, 1 + levels:1 + 2 * levels].sum()),
                      "algorithmic_bytes": int(b_alg),
                      "alg_bytes_over_peak": round(b_alg / (ms * 1e-3) / 8e12, 4) if ms > 0 else None}))
    tb.free()
    sc.close()


if __name__ == "__main__":
    main()
