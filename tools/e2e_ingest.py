#!/usr/bin/env python3
"""End-to-end timing of the CLI on real files: writes a run directory with full-size tiles
(2743 x 1571 clusters, gzip level 1) once, then times `count_well_duplicates` on it.
  python tools/e2e_ingest.py --tiles 2 --cycles 50 [--dir /tmp/wd_e2e]
"""
import argparse
import io
import os
import sys
import time
from contextlib import redirect_stdout

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from well_duplicates_amd import synth, workload, cluster_indexes          # noqa: E402
from well_duplicates_amd import count_well_duplicates as cwd              # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=2)
    ap.add_argument("--cycles", type=int, default=50)
    ap.add_argument("--dir", default="/tmp/wd_e2e")
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--extra", default="")
    ap.add_argument("--levels", type=int, default=5)
    ap.add_argument("--metric", default="-e 0 --hamming", help='e.g. "-e 2" for the reference default')
    ap.add_argument("--all-wells", action="store_true", help="every well a centre (rings from s.locs)")
    ap.add_argument("--qual-levels", type=int, default=7, help="binned qualities -> compressible planes")
    ap.add_argument("--gzip-level", type=int, default=6)
    a = ap.parse_args()
    rows, cols = workload.HISEQ4000_ROWS, workload.HISEQ4000_COLS
    spec = synth.SynthSpec(seed=2, n_clusters=rows * cols, row=cols, qual_levels=a.qual_levels)
    tiles = workload.tiles_for_stype("hiseq_x")[:a.tiles]
    marker = os.path.join(a.dir, "done_%d_%d_%d_%d" % (a.tiles, a.cycles, a.qual_levels, a.gzip_level))
    tfile = os.path.join(a.dir, "targets.list")
    if not os.path.exists(marker):
        t0 = time.time()
        os.makedirs(a.dir, exist_ok=True)
        x, y = synth.honeycomb_pixels(rows, cols)
        synth.write_run_dir(spec, a.dir, [1], tiles, list(range(a.cycles)), compresslevel=a.gzip_level,
                            slocs=synth.slocs_bytes(x, y))
        centres = cluster_indexes.sample_centres(rows * cols, 2500, 13)
        with open(tfile, "w") as fh:
            cluster_indexes.write_targets(cluster_indexes.generate(x, y, centres, 5), fh)
        open(marker, "w").write("ok")
        print("wrote run dir in %.1f s" % (time.time() - t0), file=sys.stderr)
    gz = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(a.dir) for f in fs if f.endswith(".gz"))
    argv = ["-f", tfile, "-n", "2500", "-l", str(a.levels), "-s", "hiseq_x", "-r", a.dir, "-i", "1",
            "-t", ",".join(tiles), "--cycles", "0-%d" % a.cycles, "-q", "-S",
            "--threads", str(a.threads)] + a.metric.split() + a.extra.split()
    if a.all_wells:
        argv.append("--all-wells")
    for rep in range(2):
        t0 = time.time()
        buf = io.StringIO()
        with redirect_stdout(buf):
            cwd.main(argv)
        dt = time.time() - t0
        print("run %d: %.2f s for %d tiles (%.2f s/tile), %.0f MB gz, %.0f MB raw -> %.2f GB/s raw"
              % (rep, dt, a.tiles, dt / a.tiles, gz / 1e6, a.tiles * a.cycles * rows * cols / 1e6,
                 a.tiles * a.cycles * rows * cols / dt / 1e9))
    print(buf.getvalue().splitlines()[0])


if __name__ == "__main__":
    main()
