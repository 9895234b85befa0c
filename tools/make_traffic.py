#!/usr/bin/env python3
"""profiles/traffic.json from the PMC summaries tools/pmc_collect.py left in profiles/:
HBM bytes per scan (all kernels of one wd_scan_async) per tile, which bench.py reports as
`roofline.traffic` / `other_modes.*.traffic` for the matching workload.

HBM bytes = 2 x FETCH_SIZE KiB (gfx950 tallies a 128-byte request as 64 B; calibrated for the
byte-gather pattern in profiles/r01_b_*) + WRITE_SIZE KiB, per dispatch, summed over the scan's
kernels (table builders and synthetic-data generators excluded).
"""
import glob
import json
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKIP = ("k_gen_rings", "k_transpose", "k_dense_windows", "k_synth", "k_interleave4")


def main():
    out = {"_comment": __doc__.strip().replace("\n", " ")}
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r02_*_pmc_*.json"))):
        d = json.load(open(path))
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
        out[key] = {"hbm_bytes_per_tile": total / tiles, "tiles_measured": tiles,
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
