#!/usr/bin/env python3
"""Write a run directory of N full-size tiles (GPU-generated planes, gzip -6) and run the CLI on it a
few times with WD_CLI_TIMING; for tracing one run under rocprofv3."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from well_duplicates_amd import workload  # noqa: E402
from well_duplicates_amd import count_well_duplicates as cwd  # noqa: E402

tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 8
extra = sys.argv[2:]
rows, cols = workload.HISEQ4000_ROWS, workload.HISEQ4000_COLS
centre, lvl_off, nbr = workload.honeycomb_targets(rows, cols, 2500, 5, seed=13)
orig_main = cwd.main
runs = []


def traced(argv):
    t = time.perf_counter()
    rc = orig_main(argv + extra)
    runs.append(time.perf_counter() - t)
    return rc


cwd.main = traced
bench.e2e_probe(0, tiles, rows, cols, centre, lvl_off, nbr)
print("runs:", " ".join("%.3f" % r for r in runs))
