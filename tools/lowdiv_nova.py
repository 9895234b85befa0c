#!/usr/bin/env python3
"""bench.py's configs[3] probe alone (novaseq_probe): kernel times, the interleaved layout, the low-diversity
worst case of the line walk beside the queue kernel.  Usage: lowdiv_nova.py [tiles=96]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

res = bench.novaseq_probe(0, int(sys.argv[1]) if len(sys.argv) > 1 else 96)
res.pop("cbcl_ingest", None)
print(json.dumps(res, indent=1))
