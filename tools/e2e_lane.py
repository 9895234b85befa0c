#!/usr/bin/env python3
"""A whole lane end to end: 96 full-size HiSeq X tiles x 50 cycles (4 800 .bcl.gz files, gzip -6,
binned qualities, written on the fly from GPU-generated planes), then the CLI with the reference's
default metric for several --tile-batch values, with the GPU decoder and with --host-inflate.
Usage: e2e_lane.py [tiles=96] [tile-batch,tile-batch,...] [reader threads] [lanes=1]
(further lanes are hard links of lane 1's files under their own names: the same bytes, read again)"""
import gzip
import io
import os
import shutil
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor
from contextlib import redirect_stderr, redirect_stdout

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from well_duplicates_amd import count_well_duplicates as cwd  # noqa: E402
from well_duplicates_amd import synth, workload  # noqa: E402
from well_duplicates_amd.scanner import Scanner, TileBatch  # noqa: E402

n_tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 96
batches = [int(b) for b in sys.argv[2].split(",")] if len(sys.argv) > 2 else [8, 16, 32]
cycles = 50
threads = int(sys.argv[3]) if len(sys.argv) > 3 else min(32, os.cpu_count() or 1)
n_lanes = int(sys.argv[4]) if len(sys.argv) > 4 else 1
rows, cols = workload.HISEQ4000_ROWS, workload.HISEQ4000_COLS
n = rows * cols
centre, lvl_off, nbr = workload.honeycomb_targets(rows, cols, 2500, 5, seed=13)
tiles = [str(t) for t in workload.tiles_for_stype("hiseq_x")[:n_tiles]]
root = tempfile.mkdtemp(prefix="wd_lane_")
try:
    t0 = time.perf_counter()
    ldir = os.path.join(root, "Data", "Intensities", "BaseCalls", "L001")
    for c in range(cycles):
        os.makedirs(os.path.join(ldir, "C%d.1" % (c + 1)))
    gz = [0]
    with Scanner(0) as sc:
        for first in range(0, n_tiles, 16):                     # 16 tiles of planes at a time
            part = tiles[first:first + 16]
            tb = TileBatch(sc, len(part), cycles, n)
            tb.fill_synthetic(synth.SynthSpec(seed=2, n_clusters=n, row=cols, qual_levels=7),
                              [(1, int(t)) for t in part], list(range(cycles)))

            def write(job):
                i, c = job
                data = gzip.compress(synth.bcl_file_bytes(tb.download_plane(i, c)), compresslevel=6)
                gz[0] += len(data)
                with open(os.path.join(ldir, "C%d.1" % (c + 1), "s_1_%s.bcl.gz" % part[i]), "wb") as fh:
                    fh.write(data)
            with ThreadPoolExecutor(max_workers=threads) as pool:
                list(pool.map(write, [(i, c) for i in range(len(part)) for c in range(cycles)]))
            for i, t in enumerate(part):
                with open(os.path.join(ldir, "s_1_%s.filter" % t), "wb") as fh:
                    fh.write(synth.filter_file_bytes(tb.download_filter(i)))
            tb.free()
            print("written %d tiles, %.0f s" % (first + len(part), time.perf_counter() - t0), flush=True)
    for lane in range(2, n_lanes + 1):
        ldir2 = os.path.join(root, "Data", "Intensities", "BaseCalls", "L%03d" % lane)
        for c in range(cycles):
            os.makedirs(os.path.join(ldir2, "C%d.1" % (c + 1)))
            for t in tiles:
                os.link(os.path.join(ldir, "C%d.1" % (c + 1), "s_1_%s.bcl.gz" % t),
                        os.path.join(ldir2, "C%d.1" % (c + 1), "s_%d_%s.bcl.gz" % (lane, t)))
        for t in tiles:
            os.link(os.path.join(ldir, "s_1_%s.filter" % t), os.path.join(ldir2, "s_%d_%s.filter" % (lane, t)))
    tfile = os.path.join(root, "targets.list")
    with open(tfile, "w") as fh:
        for t in range(centre.shape[0]):
            fh.write("%d\n" % centre[t])
            for l in range(lvl_off.shape[1] - 1):
                fh.write(",".join(str(int(w)) for w in nbr[lvl_off[t, l]:lvl_off[t, l + 1]]) + "\n")
    print("%d files, %.2f GB compressed, %.2f GB of planes, %d reader threads, %d cpus"
          % (n_tiles * cycles, gz[0] / 1e9, n_tiles * cycles * n / 1e9, threads, os.cpu_count()), flush=True)
    argv = ["-f", tfile, "-n", "2500", "-l", "5", "-s", "hiseq_x", "-r", root,
            "-i", ",".join(str(ln) for ln in range(1, n_lanes + 1)), "-t", ",".join(tiles),
            "--cycles", "0-%d" % cycles, "-S", "--threads", str(threads)]
    if not os.environ.get("WD_LANE_LOG"):        # WD_LANE_LOG=1: with the stderr log of duplicates, the reference's default
        argv.append("-q")
    if os.environ.get("WD_LANE_ALL_WELLS"):      # every well a centre (3 levels, rings from the run's s.locs): the dense path
        x, y = synth.honeycomb_pixels(rows, cols)
        with open(os.path.join(root, "Data", "Intensities", "s.locs"), "wb") as fh:
            fh.write(synth.slocs_bytes(x, y))
        argv = ["--all-wells", "-l", "3"] + argv[6:]

    def run(extra):
        import resource
        buf = io.StringIO()
        r0 = resource.getrusage(resource.RUSAGE_SELF)
        t1 = time.perf_counter()
        err = io.StringIO()
        with redirect_stdout(buf), redirect_stderr(err):
            # WD_LANE_EXITING=1: as `python -m well_duplicates_amd.count_well_duplicates` closes its context - the
            # process ends next, nothing is freed one by one ("fast_exit"; this tool then leaks a run's buffers)
            cwd.main(argv + extra, exiting=bool(os.environ.get("WD_LANE_EXITING")))
        for line in err.getvalue().splitlines():
            if line.startswith("[wd "):            # WD_CLI_TIMING / WD_INFLATE_STATS lines, not the duplicate log
                print(line)
        dt = time.perf_counter() - t1
        r1 = resource.getrusage(resource.RUSAGE_SELF)
        print("    run %s: %.3f s wall, cpu %.2f s user + %.2f s system"
              % (" ".join(extra), dt, r1.ru_utime - r0.ru_utime, r1.ru_stime - r0.ru_stime), flush=True)
        return dt, buf.getvalue()
    texts = set()
    # WD_LANE_VARIANTS="name:ENV=val,ENV=val;name2:..." - the same lane under other environments
    for variant in filter(None, os.environ.get("WD_LANE_VARIANTS", "").split(";")):
        name, _, assigns = variant.partition(":")
        saved = {}
        for a in filter(None, assigns.split(",")):
            key, _, val = a.partition("=")
            saved[key] = os.environ.get(key)
            os.environ[key] = val
        run(["--tile-batch", str(batches[0])])
        s, text = min(run(["--tile-batch", str(batches[0])]) for _ in range(3))
        texts.add(text)
        print("variant %-24s: %.3f s = %.2f ms per tile" % (name, s, s / n_tiles * 1e3), flush=True)
        for key, val in saved.items():
            if val is None:
                os.environ.pop(key, None)
            else:
                os.environ[key] = val
    for tbatch in batches:
        run(["--tile-batch", str(tbatch)])
        s, text = min(run(["--tile-batch", str(tbatch)]) for _ in range(2))
        texts.add(text)
        print("gpu inflate, --tile-batch %2d: %.3f s for %d lane(s) = %.2f ms per tile, %.1f GB/s of plane bytes"
              % (tbatch, s, n_lanes, s / (n_tiles * n_lanes) * 1e3, n_lanes * n_tiles * cycles * n / s / 1e9), flush=True)
    for tbatch in ([] if os.environ.get("WD_LANE_NO_HOST") else batches[-1:]):
        s, text = min(run(["--tile-batch", str(tbatch), "--host-inflate"]) for _ in range(2))
        texts.add(text)
        print("host inflate, --tile-batch %2d: %.3f s = %.2f ms per tile" % (tbatch, s, s / n_tiles * 1e3), flush=True)
    assert len(texts) == 1, "reports differ between runs"
    print(text.strip().splitlines()[-1])
finally:
    shutil.rmtree(root, ignore_errors=True)
