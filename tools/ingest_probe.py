#!/usr/bin/env python3
"""Where the time of the native loader goes: wd_load_bcl_gz on full-size planes, one thread and
many, with the library's gunzip and with zlib (fast_inflate option)."""
import argparse, os, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from well_duplicates_amd import synth, workload
from well_duplicates_amd.scanner import Scanner, TileBatch

ap = argparse.ArgumentParser()
ap.add_argument("--cycles", type=int, default=128)
ap.add_argument("--dir", default="/tmp/wd_ingest_probe")
ap.add_argument("--qual-levels", type=int, default=7)
ap.add_argument("--gzip-level", type=int, default=6)
a = ap.parse_args()
rows, cols = workload.HISEQ4000_ROWS, workload.HISEQ4000_COLS
n = rows * cols
spec = synth.SynthSpec(seed=2, n_clusters=n, row=cols, qual_levels=a.qual_levels)
if not os.path.exists(os.path.join(a.dir, "ok")):
    synth.write_run_dir(spec, a.dir, [1], ["1101"], list(range(a.cycles)), compresslevel=a.gzip_level)
    open(os.path.join(a.dir, "ok"), "w").write("1")
base = os.path.join(a.dir, "Data", "Intensities", "BaseCalls", "L001")
paths = [os.path.join(base, "C%d.1" % (c + 1), "s_1_1101.bcl.gz") for c in range(a.cycles)]
gz = sum(os.path.getsize(p) for p in paths)
sc = Scanner(0)
tb = TileBatch(sc, 1, a.cycles, n)
for fast in (1, 0, 1, 0):
    sc.set_option("fast_inflate", fast)
    for threads in (1, 4, 16, 32):
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=threads) as pool:
            list(pool.map(lambda c: sc.load_bcl_gz(paths[c], tb.plane_ptr(0, c), n), range(a.cycles)))
        dt = time.perf_counter() - t0
        print("fast_inflate=%d threads=%2d: %.3f s for %d planes (%.1f ms/plane/thread, %.2f GB/s raw, %.0f MB gz)"
              % (fast, threads, dt, a.cycles, dt / a.cycles * threads * 1e3, a.cycles * n / dt / 1e9, gz / 1e6))
print("host cpus:", os.cpu_count())
