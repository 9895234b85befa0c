#!/usr/bin/env python3
"""bench.py's end-to-end probe on its own (run directory written on the fly, CLI on it)."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from well_duplicates_amd import workload  # noqa: E402

tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rows, cols = workload.HISEQ4000_ROWS, workload.HISEQ4000_COLS
centre, lvl_off, nbr = workload.honeycomb_targets(rows, cols, 2500, 5, seed=13)
print(json.dumps(bench.e2e_probe(0, tiles, rows, cols, centre, lvl_off, nbr), indent=1))
