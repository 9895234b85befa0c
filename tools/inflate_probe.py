#!/usr/bin/env python3
"""GPU inflate (wd_load_bcl_gz_batch) beside the host loader (wd_load_bcl_gz on a thread pool) on
full-size .bcl.gz planes: seconds per batch, GB/s of plane bytes, and a check that both give the
same planes.  Planes are generated on the GPU and gzipped by a thread pool."""
import argparse
import gzip
import os
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from well_duplicates_amd import synth, workload  # noqa: E402
from well_duplicates_amd.scanner import Scanner, TileBatch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tiles", type=int, default=4)
ap.add_argument("--cycles", type=int, default=50)
ap.add_argument("--qual-levels", type=int, default=7)
ap.add_argument("--gzip-level", type=int, default=6)
ap.add_argument("--threads", default="4,16,32")
ap.add_argument("--repeat", type=int, default=3)
ap.add_argument("--chunk-mb", type=int, default=0)
ap.add_argument("--no-host", action="store_true")
ap.add_argument("--waves", type=int, default=0)
a = ap.parse_args()
rows, cols = workload.HISEQ4000_ROWS, workload.HISEQ4000_COLS
n = rows * cols
spec = synth.SynthSpec(seed=2, n_clusters=n, row=cols, qual_levels=a.qual_levels)
root = tempfile.mkdtemp(prefix="wd_inflate_probe_")
sc = Scanner(0)
if a.chunk_mb:
    sc.set_option("inflate_chunk_mb", a.chunk_mb)
sc.set_option("inflate_waves", a.waves)
tb = TileBatch(sc, a.tiles, a.cycles, n)
tiles = [(1, 1101 + i) for i in range(a.tiles)]
tb.fill_synthetic(spec, tiles, list(range(a.cycles)))
jobs = [(i, c) for i in range(a.tiles) for c in range(a.cycles)]
paths = [os.path.join(root, "t%d_c%d.bcl.gz" % j) for j in jobs]
t0 = time.perf_counter()


def write(k):
    i, c = jobs[k]
    data = gzip.compress(synth.bcl_file_bytes(tb.download_plane(i, c)), compresslevel=a.gzip_level)
    with open(paths[k], "wb") as fh:
        fh.write(data)
    return len(data)


with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as pool:
    gz = sum(pool.map(write, range(len(jobs))))
print("%d files, %.0f MB compressed (%.3f of %.0f MB) written in %.1f s; host cpus %d"
      % (len(jobs), gz / 1e6, gz / (len(jobs) * (n + 4)), len(jobs) * n / 1e6, time.perf_counter() - t0, os.cpu_count()),
      flush=True)
want = [tb.download_plane(i, c) for i, c in jobs[:3]] + [tb.download_plane(*jobs[-1])]
out = TileBatch(sc, a.tiles, a.cycles, n)
dsts = [out.plane_ptr(i, c) for i, c in jobs]


def check(label):
    got = [out.download_plane(i, c) for i, c in jobs[:3]] + [out.download_plane(*jobs[-1])]
    assert all((g == w).all() for g, w in zip(got, want)), label
    for i, c in jobs[:3] + [jobs[-1]]:
        sc.memset(out.plane_ptr(i, c), 0, n)


for threads in [int(t) for t in a.threads.split(",")]:
    for rep in range(a.repeat):
        g0, h0 = sc.get_option("inflate_files_gpu"), sc.get_option("inflate_files_host")
        t0 = time.perf_counter()
        sc.load_bcl_gz_batch(paths, dsts, n, threads=threads)
        dt = time.perf_counter() - t0
        print("gpu inflate, %2d reader threads: %.3f s = %.1f ms/tile(50 cyc), %.2f GB/s of plane bytes, %.2f GB/s "
              "compressed; decoded on the GPU %d, on the host %d"
              % (threads, dt, dt / len(jobs) * 50 * 1e3, len(jobs) * n / dt / 1e9, gz / dt / 1e9,
                 sc.get_option("inflate_files_gpu") - g0, sc.get_option("inflate_files_host") - h0), flush=True)
        check("gpu")
if not a.no_host:
    for threads in [int(t) for t in a.threads.split(",")]:
        for rep in range(min(2, a.repeat)):
            t0 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=threads) as pool:
                list(pool.map(lambda k: sc.load_bcl_gz(paths[k], dsts[k], n), range(len(jobs))))
            dt = time.perf_counter() - t0
            print("host inflate, %2d threads: %.3f s = %.1f ms/tile(50 cyc), %.2f GB/s of plane bytes"
                  % (threads, dt, dt / len(jobs) * 50 * 1e3, len(jobs) * n / dt / 1e9), flush=True)
            check("host")
for p in paths:
    os.unlink(p)
os.rmdir(root)
