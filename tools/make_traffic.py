#!/usr/bin/env python3
"""profiles/traffic.json from the PMC summaries tools/pmc_collect.py left in profiles/:
HBM bytes per scan (all kernels of one wd_scan_async) per tile, which bench.py reports as
`roofline.traffic` / `other_modes.*.traffic` for the matching workload.

These are L2-MISS bytes: FETCH_SIZE counts requests the L2 sent to the fabric, and the 256 MiB
Infinity Cache sits behind it, so where a workload re-reads lines within a launch (or a bench loop
rescans a resident lane) part of this traffic never reaches HBM.

HBM bytes = 2 x FETCH_SIZE KiB (gfx950 tallies a 128-byte request as 64 B; calibrated for the
byte-gather pattern in profiles/r01_b_*) + WRITE_SIZE KiB, per dispatch, summed over the scan's
kernels (table builders and synthetic-data generators excluded).
"""
import glob
import json
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKIP = ("k_gen_rings", "k_transpose", "k_dense_windows", "k_dense_symcheck", "k_synth", "k_interleave4")


def main():
    out = {"_comment": __doc__.strip().replace("\n", " ")}
    # a later round's run of the same case replaces the earlier one (r03_* after r02_*)
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r0*_pmc_*.json"))):
        d = json.load(open(path))
        if not d.get("probe"):
            continue                                   # (round 1's summaries have another form)
        probe = next(iter(d["probe"].values()))
        tiles = probe["tiles"]
        kernels = {k: v for k, v in d["kernels"].items() if not k.startswith(SKIP) and "hbm_bytes_per_dispatch" in v}
        if not kernels:
            continue
        total = sum(v["hbm_bytes_per_dispatch"] for v in kernels.values())
        dur = sum(v["duration_us"]["mean"] for v in kernels.values())
        key = "%s_T%d_l%d_L%d" % (probe["case"], probe["T"], probe["levels"], probe["L"])
        if probe["workload"] == "dense":
            key += "_plant%d" % probe["plant_per_64k"]
        # the compare kernel the probe's library said it launched (wd_last_kernel; runs older than that
        # entry point: the scan kernel's name as the profiler recorded it) - bench.py quotes these
        # bytes only for a run whose library reports the same kernel AND the same hash of that kernel's
        # translation unit (unit_id: wd_build_id; runs older than round 4 carry none and read as stale)
        scan = [k for k in kernels if k.startswith(("k_scan_q<", "k_scan<"))]
        kernel = probe.get("kernel") or (scan[0] if len(scan) == 1 else None)
        if kernel is None and probe["workload"] == "dense":         # round 2's chain (kDenseChainVersion 2)
            kernel = "dense chain v2, %s (k_dense_sig .. k_dense_reduce)" % (
                {"dense_eq": "equality", "dense_ham2": "Hamming", "dense_lev2": "Levenshtein <= 2"}[probe["case"]])
        out[key] = {"kernel": kernel, "unit": probe.get("unit"), "unit_id": probe.get("unit_id"),
                    "build_id": probe.get("build_id"),
                    "hbm_bytes_per_tile": total / tiles, "tiles_measured": tiles,
                    "kernel_us_per_scan": round(dur, 1), "algorithmic_bytes_per_tile": probe["algorithmic_bytes"] / tiles,
                    "kernels": {k: round(v["hbm_bytes_per_dispatch"]) for k, v in sorted(kernels.items())},
                    "source": os.path.relpath(path, REPO)}
    with open(os.path.join(REPO, "profiles", "traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    for k, v in out.items():
        if k != "_comment":
            print("%-40s %8.1f MB/tile  (alg %8.1f MB/tile)  %s" % (k, v["hbm_bytes_per_tile"] / 1e6,
                                                                    v["algorithmic_bytes_per_tile"] / 1e6, v["source"]))


if __name__ == "__main__":
    main()
