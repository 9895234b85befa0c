#!/usr/bin/env python3
"""In-process A/B of scanner options on the bench workload (interleaved rounds, one
process, HIP-event kernel times - cdna_hip_programming.md rule 24).

  python tools/tune.py --tiles 96 --variants "tpb=8,b1=4,b2=8;tpb=32,b1=4,b2=4" --rounds 5
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from well_duplicates_amd import synth, workload                     # noqa: E402
from well_duplicates_amd.scanner import Scanner, TileBatch         # noqa: E402

KEYS = {"tpb": "targets_per_block", "b1": "batch_first", "b2": "batch_next", "early": "early_exit",
        "q": "queue_kernel", "qf": "queue_first"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=96)
    ap.add_argument("--targets", type=int, default=2500)
    ap.add_argument("--levels", type=int, default=5)
    ap.add_argument("--bases", type=int, default=50)
    ap.add_argument("--mode", type=int, default=0)
    ap.add_argument("-k", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--variants", default="tpb=8,b1=4,b2=8")
    ap.add_argument("--sort-nbr", action="store_true",
                    help="experiment: sort every target's neighbour list by well index (tallies become meaningless)")
    args = ap.parse_args()
    rows, cols = workload.HISEQ4000_ROWS, workload.HISEQ4000_COLS
    n = rows * cols
    centre, lvl_off, nbr = workload.honeycomb_targets(rows, cols, args.targets, args.levels, 13)
    if args.sort_nbr:
        for t in range(centre.shape[0]):
            nbr[lvl_off[t, 0]:lvl_off[t, -1]] = np.sort(nbr[lvl_off[t, 0]:lvl_off[t, -1]])
    spec = synth.SynthSpec(seed=2, n_clusters=n, row=cols)
    tiles = [int(t) for t in workload.tiles_for_stype("hiseq_x")][:args.tiles]
    sc = Scanner(0)
    sc.set_targets(centre, lvl_off, nbr)
    tb = TileBatch(sc, len(tiles), args.bases, n)
    tb.fill_synthetic(spec, [(1, t) for t in tiles], list(range(args.bases)))
    out = sc.malloc(len(tiles) * (1 + 5 * args.levels) * 8)
    variants = [dict(kv.split("=") for kv in v.split(",")) for v in args.variants.split(";")]
    times = [[] for _ in variants]
    ref = None
    sc.set_option("profile", 1)
    for r in range(args.rounds):
        for vi, v in enumerate(variants):
            for kk, val in v.items():
                sc.set_option(KEYS[kk], int(val))
            sc.scan_async(tb.tables, len(tiles), args.bases, n, args.mode, args.k, out)
            sc.scan_status()
            sc.profile_reset()
            for _ in range(args.iters):
                sc.scan_async(tb.tables, len(tiles), args.bases, n, args.mode, args.k, out)
            ms, cnt = sc.profile_get()
            times[vi].append(ms / cnt)
            got = sc.d2h(out, len(tiles) * (1 + 5 * args.levels) * 8, np.int64)
            if ref is None:
                ref = got
            assert (got == ref).all(), "variant %s changed the counters" % v
    C = int(ref.reshape(len(tiles), -1)[:, 1:1 + args.levels].sum())
    Tv = int(ref.reshape(len(tiles), -1)[:, 0].sum())
    balg = C * (args.bases + 4) + Tv * (args.bases + 5) + 8 * (1 + 5 * args.levels) * len(tiles)
    print("compares/launch %d, valid targets %d, B_alg %.1f MB" % (C, Tv, balg / 1e6))
    for v, t in zip(variants, times):
        med, mn = float(np.median(t)), float(np.min(t))
        print("%-40s median %.4f ms  min %.4f ms  %.2f Gcmp/s  %.0f GB/s alg (%.1f%% of 8 TB/s)" % (
            ",".join("%s=%s" % kv for kv in v.items()), med, mn, C / med / 1e6, balg / med / 1e6,
            balg / med / 1e6 / 80.0))


if __name__ == "__main__":
    main()
