#!/usr/bin/env python3
"""bench.py - the headline benchmark: target x neighbour sequence compares per second.

Workload, one lane per rank (weak scaling), 2500 targets (prepare_cluster_indexes semantics, seed
13) x 5 levels, 50 bp, HiSeq-4000 tile geometry (2743 x 1571 = 4 309 253 clusters/tile), synthetic
BCL planes and filters generated on the device from the counter-based spec
(well_duplicates_amd/synth.py: 0.5 % no-calls, 70 % filter pass, 2 % planted near-duplicates):
  N = 1   BASELINE.json configs[1]: 1 lane x 96 tiles (--stype hiseq_x);
  N > 1   BASELINE.json configs[2]: N lanes x 112 tiles (--stype hiseq_4000), tile-sharded.
Rank r scans lane r+1.

A "step" is one pass of the scan path over the rank's 96 resident tiles (all kernels of
wd_scan_async) into that step's rows of the job's counter block.  For N > 1 the ranks' rows are
merged inside the timed region by one int64 all-reduce of the [steps, lanes*tiles, 1+5*levels]
block per job (ONE collective per job, as in the CLI; `--merge-every 1` merges
after every step instead).  Inputs are resident in HBM before the timed region.  One unit of work is one
performed compare = one unit of the reference report's `Wells` column
(count_well_duplicates.py:93), read back from the device's own counters.

Extra objects on the JSON line:
  roofline      HBM roofline of the dominant kernel (k_scan_q): algorithmic bytes
                B = C*(L+4) + Tv*(L+5) + 8*(1+5*levels)*tiles per launch over the
                kernel's mean duration, measured with HIP events on the launch stream around every
                n-th launch of the timed region itself (n = 4 at 20 steps: events around every launch
                cost the step 6 %), `region_ms_per_launch` = one event pair around the whole region; `traffic` = HBM bytes per launch from
                the PMC counters of the same kernel on the same workload (profiles/traffic.json,
                `traffic_source` says which run - null if that run profiled another kernel than the one
                this run launched, or another build of it: the library's per-unit source hash, wd_build_id), `l2_miss_frac` = traffic / kernel time / peak (L2-miss bytes: the
                Infinity Cache sits behind the L2); `two_lanes_alternating` = the same launches over two
                resident lanes, the control for what the cache keeps between steps.
  cpu_baseline  the C oracle (oracle/welldup_oracle.c, -O3, OpenMP over tiles) on a bounded
                sample of the same tiles, on this box's host cores.
  e2e           real files through the CLI path: a run directory of full-size tiles is written on
                the fly (.bcl.gz, gzip -6, binned qualities), then count_well_duplicates runs on
                it with the reference's default metric: seconds per tile, plane bytes per second.
  other_modes   kernel times of the other compare modes / layouts / the dense all-centres path;
                `alg_bytes_over_peak` is algorithmic bytes / time / 8 TB/s - it can exceed 1 where a
                lazy kernel never fetches what the byte model charges - and `l2_miss_frac` is the
                counter-based fraction beside it.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--tiles", type=int, default=None,
                    help="tiles per rank (one lane); default: all tiles of --stype (96 / 112)")
    ap.add_argument("--stype", default=None, choices=["hiseq_x", "hiseq_4000"],
                    help="lane layout: hiseq_x = 96 tiles (BASELINE configs[1], default at N = 1), "
                         "hiseq_4000 = 112 (configs[2], default at N > 1)")
    ap.add_argument("--targets", type=int, default=2500)
    ap.add_argument("--levels", type=int, default=5)
    ap.add_argument("--bases", type=int, default=50)
    ap.add_argument("--mode", default="eq", choices=["eq", "hamming", "levenshtein"])
    ap.add_argument("-k", type=int, default=0, help="distance threshold (hamming / levenshtein)")
    ap.add_argument("--no-early-exit", action="store_true",
                    help="full gather: read all L bytes of every neighbour")
    ap.add_argument("--cpu-tiles", type=int, default=16,
                    help="tiles in the CPU-baseline sample (one thread per tile, up to the CPUs this process may use)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=20)
    ap.add_argument("--interleaved-tiles", type=int, default=96,
                    help="tiles of the interleaved-layout measurement under other_modes; 0 = skip")
    ap.add_argument("--e2e-tiles", type=int, default=16,
                    help="full-size tiles of the end-to-end (real files) measurement; 0 = skip")
    ap.add_argument("--novaseq-tiles", type=int, default=96,
                    help="tiles of the BASELINE configs[3] shape (NovaSeq tiles, 10000 targets x 7 levels) reported "
                         "under other_modes; 0 = skip")
    ap.add_argument("--dense-tiles", type=int, default=8,
                    help="tiles of the all-centres probe (BASELINE configs[4] shape: every well a centre, "
                         "3 levels, 150 bp) reported under other_modes; 0 = skip")
    ap.add_argument("--dense-tiles-large", type=int, default=96,
                    help="a second all-centres probe on this many tiles - a lane (equality and Levenshtein <= 2); 0 = skip")
    ap.add_argument("--option", action="append", default=[], help="name=value scanner option")
    ap.add_argument("--merge-every", type=int, default=0,
                    help="steps merged by one all-reduce (N > 1); 0 = one merge per job (all timed steps)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1: nccl = RCCL over xGMI (default); gloo = "
                         "rehearsal of the multi-rank logic with several ranks on ONE GPU")
    ap.add_argument("--master-port", type=int, default=0,
                    help="rendezvous port when bench.py starts its own ranks (0 = a free one)")
    ap.add_argument("--first-lane", type=int, default=1, help="rank r scans lane FIRST_LANE + r (x LANES_PER_RANK)")
    ap.add_argument("--lanes-per-rank", type=int, default=1,
                    help="lanes resident on one rank's GPU and scanned per step (default 1: BASELINE's one lane per "
                         "GPU).  8 on one rank = the whole of BASELINE configs[2]'s input (8 lanes x 112 tiles, 193 GB) "
                         "on one MI355X; 2 on 4 ranks = the same flowcell over four processes")
    ap.add_argument("--check-tiles", type=int, default=0,
                    help="after the timed region, rank 0 checks this many tiles, sampled over ALL ranks' rows of the "
                         "merged block, against the C oracle (the planes are regenerated on the host from the spec)")
    ap.add_argument("--dump-block", default=None,
                    help="rank 0 saves the last step's merged [ranks*tiles, 1+5*levels] counter block here (.npy)")
    ap.add_argument("--dry-run", action="store_true",
                    help="print the launcher's command line (JSON) instead of running it; touches no GPU")
    return ap.parse_args(argv)


def launcher_argv(gpus, argv, port):
    """The child command `bench.py --gpus N` starts when it was not started under torchrun: one
    rank per GPU of this node (the driver's own command line for N > 1, CONTRACT: --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a CHILD process (never an
    exec, and before this process has made any GPU / torch.cuda call - it never makes one), relay
    the child's output, leave with its exit code."""
    import subprocess
    cmd = launcher_argv(args.gpus, [a for a in argv if a != "--dry-run"], args.master_port or free_port())
    if args.dry_run:
        print(json.dumps({"launcher": cmd}))
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    env["WD_BENCH_LAUNCHER"] = "bench.py"
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True, bufsize=1)
    try:
        for line in proc.stdout:
            sys.stdout.write(line)
            sys.stdout.flush()
        return proc.wait()
    except BaseException:
        proc.terminate()          # exactly the child we started
        try:
            proc.wait(timeout=30)
        except subprocess.TimeoutExpired:
            proc.kill()
        raise


ACHIEVABLE_HBM_FRAC = 0.79     # MI355X_MICROARCH.md: ~6.29 TB/s of the 8 TB/s are achievable by a streaming read


def traffic_lookup(key, tiles, kernel):
    """Counter bytes per scan of `tiles` tiles (profiles/traffic.json, made by tools/make_traffic.py from
    the rocprofv3 --pmc passes) -> (bytes or None, source or reason).  The counter run is quoted only
    if it profiled the kernel this run's library says it launched (wd_last_kernel): a profile of
    another instantiation is stale evidence, and then `traffic` is null with the reason."""
    tpath = os.path.join(REPO, "profiles", "traffic.json")
    try:
        entry = json.load(open(tpath)).get(key)
    except Exception as e:                                    # noqa: BLE001
        return None, "profiles/traffic.json unreadable: %s" % e
    if not entry:
        return None, "no counter run for %s in profiles/traffic.json" % key
    if entry.get("kernel") != kernel:
        return None, "stale counter run: %s profiled %s, this run launched %s" % (entry["source"], entry.get("kernel"), kernel)
    # ... and only if that kernel's CODE is the code that was profiled: the library carries a hash of the
    # sources of each translation unit (wd_build_id), the counter run recorded the one it saw
    from well_duplicates_amd import _lib
    unit = _lib.unit_of_kernel(kernel)
    here = _lib.build_ids().get(unit)
    if entry.get("unit_id") != here:
        return None, "stale counter run: %s profiled unit %s at %s, this library's is %s" % (
            entry["source"], unit, entry.get("unit_id"), here)
    src = "%s [%s, %s]" % (entry["source"], key, kernel)
    if entry["tiles_measured"] != tiles:
        src += ", per-tile bytes of the %d-tile counter run x %d tiles" % (entry["tiles_measured"], tiles)
    return int(entry["hbm_bytes_per_tile"] * tiles), src


def traffic_fields(traffic, src, ms):
    """`traffic` = 2 x FETCH_SIZE + WRITE_SIZE: bytes the L2 asked the fabric for.  The Infinity Cache
    (256 MiB) sits behind the L2, so these are L2-MISS bytes - an upper bound of the HBM bytes - and
    the fraction is named for what it is; above the achievable HBM rate it proves only that part of
    the misses were served by the Infinity Cache."""
    frac = None if not (traffic and ms > 0) else round(traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    out = {"traffic": traffic, "l2_miss_frac": frac, "traffic_source": src}
    if frac is not None and frac > ACHIEVABLE_HBM_FRAC:
        out["traffic_note"] = ("L2-miss bytes at %.2f of peak exceed what HBM can deliver (%.2f): part of them are "
                               "Infinity-Cache hits; not an HBM figure" % (frac, ACHIEVABLE_HBM_FRAC))
    return out


def with_traffic(res, key, tiles, ms, kernel):
    """Add `kernel`, `traffic`, `l2_miss_frac`, `traffic_source` to a per-mode result dict."""
    traffic, src = traffic_lookup(key, tiles, kernel)
    res["kernel"] = kernel
    res.update(traffic_fields(traffic, src, ms))
    return res


def dense_probe(device, n_tiles, rows, cols, levels=3, bases=150, modes=("levenshtein_k2", "hamming_k2", "equality")):
    """Kernel time of the dense path (scan_dense.inc) on n_tiles full-size tiles."""
    import numpy as np
    from well_duplicates_amd import synth
    from well_duplicates_amd.scanner import Scanner, TileBatch, MODE_EQ, MODE_HAMMING, MODE_LEVENSHTEIN
    n = rows * cols
    x, y = synth.honeycomb_pixels(rows, cols)
    sc = Scanner(device)
    try:
        t0 = time.perf_counter()
        T, P = sc.targets_from_coords(x, y, None, levels=levels)
        gen_s = time.perf_counter() - t0
        spec = synth.SynthSpec(seed=5, n_clusters=n, row=cols)
        tb = TileBatch(sc, n_tiles, bases, n)
        tb.fill_synthetic(spec, [(1, 1101 + i) for i in range(n_tiles)], list(range(bases)))
        ncnt = 1 + 5 * levels
        out = sc.malloc(n_tiles * ncnt * 8)
        out1 = sc.malloc(ncnt * 8)
        sc.scan_async(tb.tables, n_tiles, bases, n, MODE_EQ, 0, out)      # builds the tables
        sc.set_option("profile", 1)
        b_dense = n_tiles * (n * bases + 4 * n * (1 + P / T) + n)             # SURVEY.md 8d, dense form
        res = {"workload": "%d tiles x %d centres x %.1f neighbours, %d levels, %d bp, 2 %% planted"
                           % (n_tiles, T, P / T, levels, bases),
               "algorithmic_bytes": int(b_dense), "ring_generator_s": round(gen_s, 3),
               "window_groups": sc.get_option("dense_window_groups"), "groups": (T + 63) // 64,
               # 1: the neighbour relation is symmetric, every pair is compared from its lower well only
               "pairs_from_one_end": sc.get_option("dense_sym_on"),
               # this box's rate for a kernel that only reads (roofline.stream_read): k_dense_pack and k_dense_sig
               # stream 150 planes per tile, 0.8 of the chain's time
               "stream_read_gbs": round(sc.stream_read_gbs(tb.d_planes, tb.plane_bytes - tb.plane_bytes % 16, passes=3), 1)}
        # the reference's default metric (Levenshtein <= 2), Hamming <= 2, then equality for the counters
        for name, mode, k, case in (("levenshtein_k2", MODE_LEVENSHTEIN, 2, "dense_lev2"),
                                    ("hamming_k2", MODE_HAMMING, 2, "dense_ham2"), ("equality", MODE_EQ, 0, "dense_eq")):
            if name not in modes:
                continue
            sc.scan_async(tb.tables, n_tiles, bases, n, mode, k, out)
            sc.profile_reset()
            for _ in range(4):
                sc.scan_async(tb.tables, n_tiles, bases, n, mode, k, out)
            ms, launches = sc.profile_get()
            ms /= max(1, launches)
            blk = sc.d2h(out, n_tiles * ncnt * 8, np.int64).reshape(n_tiles, ncnt)
            dense_name = sc.last_kernel()
            # no figure without a check: the queue kernel (the one the golden fixtures pin) must give
            # the same tally block on one of the timed tiles
            sc.set_option("dense_kernel", 0)
            sc.scan_async(tb.tables, 1, bases, n, mode, k, out1)
            ref = sc.d2h(out1, ncnt * 8, np.int64)
            referee = sc.last_kernel()
            sc.set_option("dense_kernel", -1)
            if not (ref == blk[0]).all():
                raise SystemExit("bench: dense path and queue kernel disagree on tile 0 (%s)" % name)
            compares = int(blk[:, 1:1 + levels].sum())
            res["checked_against"] = "%s on tile 0 of the timed tiles: tally block identical, every mode" % referee
            res[name] = with_traffic(
                {"kernel_ms": round(ms, 4), "ms_per_tile": round(ms / n_tiles, 4),
                 "compares_per_s": round(compares / (ms * 1e-3), 1),
                 "duplicates_found": int(blk[:, 1 + levels:1 + 2 * levels].sum()),
                 "alg_bytes_over_peak": round(b_dense / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                 # the honest floor of this path: every plane byte and filter byte once (the explicit
                 # neighbour table, half of the byte model, is never read by the window groups)
                 "planes_only_frac": round(n_tiles * (n * bases + n) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                "%s_T%d_l%d_L%d_plant%d" % (case, T, levels, bases, spec.plant_per_64k), n_tiles, ms, dense_name)
        tb.free()
        return res
    finally:
        sc.close()


def novaseq_probe(device, n_tiles, levels=7, targets=10000, bases=50):
    """BASELINE configs[3] shape on one GPU: NovaSeq-style tiles (4 091 904 wells), 10 000 sampled
    targets x 7 rings (device ring generator; the reference's own stops at 5), equality and the
    reference's default Levenshtein <= 2.  (The .cbcl ingest of this config is exercised by
    tests/test_gpu_novaseq.py; here the planes are resident.)"""
    import numpy as np
    from well_duplicates_amd import cluster_indexes, synth, workload
    from well_duplicates_amd.scanner import Scanner, TileBatch, MODE_EQ, MODE_LEVENSHTEIN
    rows, cols = workload.NOVASEQ_ROWS, workload.NOVASEQ_COLS
    n = rows * cols
    x, y = synth.honeycomb_pixels(rows, cols)
    with Scanner(device) as sc:
        t0 = time.perf_counter()
        T, P = sc.targets_from_coords(x, y, cluster_indexes.sample_centres(n, targets, 13), levels=levels,
                                      max_dists=cluster_indexes.max_dists_for(levels))
        gen_s = time.perf_counter() - t0
        spec = synth.SynthSpec(seed=4, n_clusters=n, row=cols)
        tile_ids = workload.tiles_for_stype(workload.NOVASEQ_STYPE)[:n_tiles]
        tb = TileBatch(sc, n_tiles, bases, n)
        tb.fill_synthetic(spec, [(1, int(t)) for t in tile_ids], list(range(bases)))
        ncnt = 1 + 5 * levels
        out = sc.malloc(n_tiles * ncnt * 8)
        sc.set_option("profile", 1)
        res = {"workload": "%d of a lane's 936 tiles x %d wells, %d targets x %d rings (%.1f neighbours), %d bp"
                           % (n_tiles, n, T, levels, P / T, bases), "ring_generator_s": round(gen_s, 3)}
        for name, mode, k, case in (("equality", MODE_EQ, 0, "eq"), ("levenshtein_k2", MODE_LEVENSHTEIN, 2, "lev2")):
            sc.scan_async(tb.tables, n_tiles, bases, n, mode, k, out)
            sc.profile_reset()
            for _ in range(5):
                sc.scan_async(tb.tables, n_tiles, bases, n, mode, k, out)
            ms, launches = sc.profile_get()
            ms /= max(1, launches)
            blk = sc.d2h(out, n_tiles * ncnt * 8, np.int64).reshape(n_tiles, ncnt)
            C, Tv = int(blk[:, 1:1 + levels].sum()), int(blk[:, 0].sum())
            b_alg = C * (bases + 4) + Tv * (bases + 5) + 8 * ncnt * n_tiles
            res[name] = with_traffic(
                {"kernel_ms": round(ms, 4), "compares": C, "compares_per_s": round(C / (ms * 1e-3), 1),
                 "algorithmic_bytes": b_alg,
                 "alg_bytes_over_peak": round(b_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                "novaseq_%s_T%d_l%d_L%d" % (case, T, levels, bases), n_tiles, ms, sc.last_kernel())
        # the same tiles with their cycles interleaved by four: the line walk reads a dword per pair and round
        # (k_scan_lines<.., 4>), a line of 32 wells x 4 cycles once per group instead of 128 wells once per cycle
        try:
            il = TileBatch(sc, n_tiles, bases, n, interleave=4)
            il.fill_synthetic(spec, [(1, int(t)) for t in tile_ids], list(range(bases)))
            out2 = sc.malloc(n_tiles * ncnt * 8)
            res["interleaved_by_4"] = {}
            for name, mode, k, case in (("equality", MODE_EQ, 0, "il"), ("levenshtein_k2", MODE_LEVENSHTEIN, 2, "il_lev2")):
                sc.set_option("well_stride", 4)
                sc.scan_async(il.tables, n_tiles, bases, n, mode, k, out2)
                sc.profile_reset()
                for _ in range(5):
                    sc.scan_async(il.tables, n_tiles, bases, n, mode, k, out2)
                ms, launches = sc.profile_get()
                kern = sc.last_kernel()
                sc.set_option("well_stride", 1)
                ms /= max(1, launches)
                sc.scan_async(tb.tables, n_tiles, bases, n, mode, k, out)          # the plane layout's counters, to compare
                sc.scan_status()
                blk = sc.d2h(out, n_tiles * ncnt * 8, np.int64).reshape(n_tiles, ncnt)
                blk2 = sc.d2h(out2, n_tiles * ncnt * 8, np.int64).reshape(n_tiles, ncnt)
                C, Tv = int(blk2[:, 1:1 + levels].sum()), int(blk2[:, 0].sum())
                b_alg = C * (bases + 4) + Tv * (bases + 5) + 8 * ncnt * n_tiles
                res["interleaved_by_4"][name] = with_traffic(
                    {"kernel_ms": round(ms, 4), "compares_per_s": round(C / (ms * 1e-3), 1),
                     "alg_bytes_over_peak": round(b_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "same_counters_as_plane_layout": bool((blk == blk2).all())},
                    "novaseq_%s_T%d_l%d_L%d" % (case, T, levels, bases), n_tiles, ms, kern)
            sc.free(out2)
            il.free()
        except Exception as e:          # noqa: BLE001 - a measurement, not the scan
            sc.set_option("well_stride", 1)
            res["interleaved_by_4"] = {"error": repr(e)}
        # the lazy gather's worst input on this shape: every read equal (all no-calls), so that no pair dies
        # early - the line walk (steps finished in place: kLwInPlace) beside the queue kernel on the same tiles
        try:
            n_ld = n_tiles                      # (the whole probe: a few tiles do not fill the chip)
            flat = synth.SynthSpec(seed=4, n_clusters=n, row=cols, nocall_per_64k=65536)
            low = {"what": "%d tiles, every read equal: each of a target's %.0f neighbours is a duplicate and is read to "
                           "its last cycle" % (n_ld, P / T)}
            out3 = sc.malloc(n_ld * ncnt * 8)
            for stride, key in ((1, "planes"), (4, "interleaved_by_4")):
                ld = TileBatch(sc, n_ld, bases, n, interleave=stride)
                ld.fill_synthetic(flat, [(1, int(t)) for t in tile_ids[:n_ld]], list(range(bases)))
                sc.set_option("well_stride", stride)
                for name, mode, k in (("equality", MODE_EQ, 0), ("levenshtein_k2", MODE_LEVENSHTEIN, 2)):
                    for walk, wname in ((1, "line_walk"), (0, "queue_kernel")):
                        sc.set_option("line_walk", walk)
                        sc.scan_async(ld.tables, n_ld, bases, n, mode, k, out3)
                        sc.profile_reset()
                        for _ in range(3):
                            sc.scan_async(ld.tables, n_ld, bases, n, mode, k, out3)
                        w_ms, w_n = sc.profile_get()
                        low["%s_%s_%s_us_per_tile" % (key, name, wname)] = round(w_ms / max(1, w_n) / n_ld * 1e3, 2)
                    a_, b_ = low["%s_%s_line_walk_us_per_tile" % (key, name)], low["%s_%s_queue_kernel_us_per_tile" % (key, name)]
                    low["%s_%s_line_walk_over_queue" % (key, name)] = round(a_ / b_, 3) if b_ else None
                sc.set_option("well_stride", 1)
                sc.set_option("line_walk", -1)
                sc.scan_status()
                blk3 = sc.d2h(out3, n_ld * ncnt * 8, np.int64).reshape(n_ld, ncnt)
                low["all_duplicates"] = bool((blk3[:, 1 + levels:1 + 2 * levels] == blk3[:, 1:1 + levels]).all())
                ld.free()
            sc.free(out3)
            res["low_diversity_worst_case"] = low
        except Exception as e:          # noqa: BLE001 - a measurement, not the scan
            sc.set_option("well_stride", 1)
            sc.set_option("line_walk", -1)
            res["low_diversity_worst_case"] = {"error": repr(e)}
        # the .cbcl side of this config: 8 tiles x `bases` cycles as NovaSeq writes them (one file per
        # cycle and surface, one gzip block per tile: 2 wells per byte, 2 quality bits, excluded wells
        # left out), through the GPU decoder (wd_load_cbcl_batch) and through the host loader
        try:
            res["cbcl_ingest"] = cbcl_ingest_probe(sc, tb, spec, [int(t) for t in tile_ids[:min(8, n_tiles)]], bases, n)
        except Exception as e:          # noqa: BLE001 - a measurement, not the scan
            res["cbcl_ingest"] = {"error": repr(e)}
        tb.free()
        return res


def cbcl_ingest_probe(sc, tb, spec, tile_ids, cycles, n):
    import gzip
    import shutil
    import struct
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    import numpy as np
    from well_duplicates_amd import synth
    from well_duplicates_amd.scanner import TileBatch
    threads = min(32, os.cpu_count() or 1)
    root = tempfile.mkdtemp(prefix="wd_cbcl_")
    try:
        filters = {t: synth.filter_bytes(spec, 1, t) for t in tile_ids}
        keep = {t: (filters[t] & 1) == 1 for t in tile_ids}
        surfaces = sorted({str(t)[0] for t in tile_ids})
        gz_total = [0]

        def write(c):
            for sf in surfaces:
                mine = [t for t in tile_ids if str(t)[0] == sf]
                blocks, table = [], b""
                for t in mine:
                    plane = tb.download_plane(tile_ids.index(t), c)
                    nib = np.where(plane == 0, 0, (plane & 3) | (((plane >> 2) % 3 + 1) << 2)).astype(np.uint8)[keep[t]]
                    cnt = nib.shape[0]
                    if cnt % 2:
                        nib = np.concatenate([nib, np.zeros(1, np.uint8)])
                    packed = (nib[0::2] | (nib[1::2] << 4)).astype(np.uint8).tobytes()
                    comp = gzip.compress(packed, compresslevel=6)
                    blocks.append(comp)
                    table += struct.pack("<IIII", t, cnt, len(packed), len(comp))
                    gz_total[0] += len(comp)
                head = struct.pack("<HIBBI", 1, 5681, 2, 2, 4) + b"".join(struct.pack("<II", i, i) for i in range(4))
                head += struct.pack("<I", len(mine)) + table + b"\x01"
                with open(os.path.join(root, "C%d_%s.cbcl" % (c, sf)), "wb") as fh:
                    fh.write(head + b"\0" * (5681 - len(head)) + b"".join(blocks))
        with ThreadPoolExecutor(max_workers=threads) as pool:
            list(pool.map(write, range(cycles)))
        out = TileBatch(sc, len(tile_ids), cycles, n)
        for i, t in enumerate(tile_ids):
            sc.h2d(out.filter_ptr(i), filters[t])
        entries = [(os.path.join(root, "C%d_%s.cbcl" % (c, str(t)[0])), t, out.filter_ptr(i), out.plane_ptr(i, c))
                   for i, t in enumerate(tile_ids) for c in range(cycles)]

        def gpu():
            t1 = time.perf_counter()
            sc.load_cbcl_batch(entries, n, threads=threads)
            return time.perf_counter() - t1

        def host():
            t1 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=threads) as pool:
                list(pool.map(lambda e: sc.load_cbcl_tile(e[0], e[1], e[2], n, e[3]), entries))
            return time.perf_counter() - t1
        gpu()
        g0 = sc.get_option("inflate_files_gpu")
        gpu_s = min(gpu() for _ in range(3))
        on_gpu = (sc.get_option("inflate_files_gpu") - g0) // 3
        check = out.download_plane(len(tile_ids) - 1, cycles - 1)
        host()
        host_s = min(host() for _ in range(2))
        same = bool((out.download_plane(len(tile_ids) - 1, cycles - 1) == check).all())
        want = tb.download_plane(len(tile_ids) - 1, cycles - 1)
        k = keep[tile_ids[-1]]
        codes_ok = bool((np.where(check[k] == 0, 4, check[k] & 3) == np.where(want[k] == 0, 4, want[k] & 3)).all()
                        and (check[~k] == 0).all())
        plane_bytes = len(entries) * n
        out.free()
        return {"tiles": len(tile_ids), "blocks": len(entries), "decoded_on_gpu": int(on_gpu), "gz_bytes": gz_total[0],
                "plane_bytes": plane_bytes, "gpu_inflate_seconds": round(gpu_s, 4),
                "gpu_inflate_plane_gb_per_s": round(plane_bytes / gpu_s / 1e9, 2),
                "gpu_kernel_ms_per_block": round(sc.get_option("inflate_us_per_file") / 1e3, 2),
                "host_inflate_seconds": round(host_s, 4), "host_inflate_plane_gb_per_s": round(plane_bytes / host_s / 1e9, 2),
                "threads": threads, "same_planes": same, "base_codes_as_generated": codes_ok}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def e2e_probe(device, n_tiles, rows, cols, centre, lvl_off, nbr, cycles=50, threads=None):
    """End to end on real files: a run directory of n_tiles full-size tiles (planes generated on the
    GPU with binned qualities, gzip -6 .bcl.gz files written by a thread pool) and the targets file,
    then this package's count_well_duplicates CLI with the reference's default metric (-e 2,
    Levenshtein), -q -S: everything from process-level file reads to the printed report."""
    import gzip
    import io
    import shutil
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    from contextlib import redirect_stdout
    from well_duplicates_amd import synth, workload
    from well_duplicates_amd import count_well_duplicates as cwd
    from well_duplicates_amd.scanner import Scanner, TileBatch
    n = rows * cols
    threads = threads or cwd.default_threads()
    spec = synth.SynthSpec(seed=2, n_clusters=n, row=cols, qual_levels=7)
    tiles = [str(t) for t in workload.tiles_for_stype("hiseq_x")[:n_tiles]]
    root = tempfile.mkdtemp(prefix="wd_e2e_")
    try:
        t0 = time.perf_counter()
        ldir = os.path.join(root, "Data", "Intensities", "BaseCalls", "L001")
        for c in range(cycles):
            os.makedirs(os.path.join(ldir, "C%d.1" % (c + 1)))
        with Scanner(device) as sc:
            tb = TileBatch(sc, n_tiles, cycles, n)
            tb.fill_synthetic(spec, [(1, int(t)) for t in tiles], list(range(cycles)))
            gz_bytes = [0]

            def write(job):
                i, c = job
                data = gzip.compress(synth.bcl_file_bytes(tb.download_plane(i, c)), compresslevel=6)
                gz_bytes[0] += len(data)
                with open(os.path.join(ldir, "C%d.1" % (c + 1), "s_1_%s.bcl.gz" % tiles[i]), "wb") as fh:
                    fh.write(data)
            with ThreadPoolExecutor(max_workers=threads) as pool:
                list(pool.map(write, [(i, c) for i in range(n_tiles) for c in range(cycles)]))
            for i, t in enumerate(tiles):
                with open(os.path.join(ldir, "s_1_%s.filter" % t), "wb") as fh:
                    fh.write(synth.filter_file_bytes(tb.download_filter(i)))
            tb.free()
        tfile = os.path.join(root, "targets.list")
        with open(tfile, "w") as fh:
            for t in range(centre.shape[0]):
                fh.write("%d\n" % centre[t])
                for l in range(lvl_off.shape[1] - 1):
                    fh.write(",".join(str(int(w)) for w in nbr[lvl_off[t, l]:lvl_off[t, l + 1]]) + "\n")
        write_s = time.perf_counter() - t0
        levels = lvl_off.shape[1] - 1
        argv = ["-f", tfile, "-n", str(centre.shape[0]), "-l", str(levels), "-s", "hiseq_x", "-r", root, "-i", "1",
                "-t", ",".join(tiles), "--cycles", "0-%d" % cycles, "-q", "-S", "--threads", str(threads),
                "--device", str(device)]                      # --tile-batch: the CLI's default

        def run(extra):
            buf = io.StringIO()
            t1 = time.perf_counter()
            with redirect_stdout(buf):
                cwd.main(argv + extra)
            return time.perf_counter() - t1, buf.getvalue()
        run([])                                               # page cache, allocator, first-use costs
        best, text = min(run([]) for _ in range(2))
        serial, text_s = min(run(["--serial-ingest"]) for _ in range(2))
        run(["--host-inflate"])
        host, text_h = min(run(["--host-inflate"]) for _ in range(2))
        inter, text_i = min(run(["--layout", "planes"]) for _ in range(2))       # (the default is --layout auto = interleaved here)
        # ... and as the reference runs by default: with the three stderr lines per duplicate (:260-262)
        import contextlib
        argv_log = [a for a in argv if a != "-q"]

        def run_logged():
            buf, err = io.StringIO(), io.StringIO()
            t1 = time.perf_counter()
            with redirect_stdout(buf), contextlib.redirect_stderr(err):
                cwd.main(argv_log)
            return time.perf_counter() - t1, buf.getvalue(), err.getvalue().count("edit distance:")
        run_logged()
        logged, text_l, dup_lines = min(run_logged() for _ in range(2))
        plane_bytes = n_tiles * cycles * n
        # the loaders alone on the same files: every .bcl.gz of the run directory into HBM, by the GPU
        # decoder (wd_load_bcl_gz_batch) and by the host threads (wd_load_bcl_gz), no scan, no report
        with Scanner(device) as sc:
            tb = TileBatch(sc, n_tiles, cycles, n)
            jobs = [(i, c) for i in range(n_tiles) for c in range(cycles)]
            paths = [os.path.join(ldir, "C%d.1" % (c + 1), "s_1_%s.bcl.gz" % tiles[i]) for i, c in jobs]
            dsts = [tb.plane_ptr(i, c) for i, c in jobs]

            def gpu_load():
                t1 = time.perf_counter()
                sc.load_bcl_gz_batch(paths, dsts, n, threads=threads)
                return time.perf_counter() - t1

            def host_load():
                t1 = time.perf_counter()
                with ThreadPoolExecutor(max_workers=threads) as pool:
                    list(pool.map(lambda k: sc.load_bcl_gz(paths[k], dsts[k], n), range(len(jobs))))
                return time.perf_counter() - t1
            gpu_load()
            g0 = sc.get_option("inflate_files_gpu")
            gpu_s = min(gpu_load() for _ in range(3))
            on_gpu = (sc.get_option("inflate_files_gpu") - g0) // 3
            host_load()
            host_s = min(host_load() for _ in range(2))
            ingest = {"files": len(jobs), "decoded_on_gpu": int(on_gpu),
                      "gpu_inflate_seconds": round(gpu_s, 4), "gpu_inflate_plane_gb_per_s": round(plane_bytes / gpu_s / 1e9, 2),
                      "gpu_kernel_ms_per_file": round(sc.get_option("inflate_us_per_file") / 1e3, 2),
                      "host_inflate_seconds": round(host_s, 4),
                      "host_inflate_plane_gb_per_s": round(plane_bytes / host_s / 1e9, 2), "threads": threads,
                      "note": "whole-batch wall time, files in the page cache; the GPU decoder works on all files of a launch "
                              "at once, so its time per batch is the kernel's time per file plus "
                              "the time to read and copy the compressed bytes"}
            tb.free()
        return {"what": "count_well_duplicates CLI (-e 2 Levenshtein, %d targets x %d levels, -q -S) on %d full-size "
                        "tiles x %d cycles of .bcl.gz files (gzip -6, 7 quality bins), warm page cache; the CLI's "
                        "default settings (batches of about 512 files, %d reader threads)"
                        % (centre.shape[0], levels, n_tiles, cycles, threads),
                "tiles": n_tiles, "files": n_tiles * cycles, "gz_bytes": gz_bytes[0], "plane_bytes": plane_bytes,
                "threads": threads, "seconds": round(best, 4), "s_per_tile": round(best / n_tiles, 4),
                "plane_gb_per_s": round(plane_bytes / best / 1e9, 3),
                "serial_ingest_seconds": round(serial, 4),
                "overlap_gain": round(serial / best, 3) if best > 0 else None,
                "inflate": "GPU (csrc/gpu_inflate.inc: four or eight waves decode one .bcl.gz file, all files of a batch at once); host threads only read the files",
                "host_inflate_seconds": round(host, 4),
                "gpu_inflate_gain": round(host / best, 3) if best > 0 else None,
                "ingest_only": ingest,
                "resident_layout": "interleaved by four (--layout auto: -e 2 on sampled targets of a .bcl.gz run)",
                "planes_layout_seconds": round(inter, 4),
                "with_duplicate_log_seconds": round(logged, 4), "duplicates_logged": dup_lines,
                "same_report": text == text_s == text_i == text_h == text_l, "run_dir_write_s": round(write_s, 1),
                "reference_s_per_tile": 7.9,
                "reference_note": "unmodified reference, 1 core, same geometry, --hamming -e 0 (BASELINE.md; measured in "
                                  "the build container, not on this box)"}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not under torchrun: this process becomes the launcher and never touches the GPU
        sys.exit(launch_ranks(args, argv))
    if args.dry_run:
        print(json.dumps({"launcher": None, "note": "runs in this process (N = 1 or already under torchrun)"}))
        return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world

    import torch
    import torch.distributed as dist
    rehearsal = args.backend == "gloo"
    if rehearsal:
        local_rank = 0                       # every rank shares GPU 0; collectives go through the CPU
    torch.cuda.set_device(local_rank)
    # WD_BENCH_FORCE_DIST=1 exercises the RCCL code path with a single rank (1-GPU boxes)
    use_dist = world > 1 or os.environ.get("WD_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # proof that the process group holds N ranks: its own size, an all-reduce of ones through the
    # backend the merge uses, and which device each rank sits on
    ranks_seen = None
    if use_dist:
        ones = torch.ones(1, dtype=torch.int64, device="cpu" if rehearsal else torch.device("cuda", local_rank))
        dist.all_reduce(ones)
        where = [None] * dist.get_world_size()
        dist.all_gather_object(where, {"rank": rank, "device": local_rank,
                                       "gpu": torch.cuda.get_device_name(local_rank)})
        ranks_seen = {"rccl_world": dist.get_world_size(), "allreduce_of_ones": int(ones.item()),
                      "backend": dist.get_backend(), "launched_by": os.environ.get("WD_BENCH_LAUNCHER", "torchrun"),
                      "ranks": where}

    from well_duplicates_amd import dist as wdist
    from well_duplicates_amd import synth, workload
    from well_duplicates_amd.scanner import Scanner, TileBatch, MODE_EQ, MODE_HAMMING, MODE_LEVENSHTEIN

    mode = {"eq": MODE_EQ, "hamming": MODE_HAMMING, "levenshtein": MODE_LEVENSHTEIN}[args.mode]
    k = 0 if args.mode == "eq" else args.k
    rows, cols = workload.HISEQ4000_ROWS, workload.HISEQ4000_COLS
    n_clusters = rows * cols
    L, levels, T = args.bases, args.levels, args.targets
    ncnt = 1 + 5 * levels

    # ---- inputs: targets (replicated), this rank's lane of tiles (resident) -------------
    t0 = time.time()
    centre, lvl_off, nbr = workload.honeycomb_targets(rows, cols, T, levels, seed=13)
    spec = synth.SynthSpec(seed=2, n_clusters=n_clusters, row=cols)
    lpr = max(1, args.lanes_per_rank)
    lane = args.first_lane + rank * lpr
    if args.stype is None:                                    # the layout BASELINE names for this N
        args.stype = "hiseq_x" if world * lpr == 1 else "hiseq_4000"
    tile_ids = [int(t) for t in workload.tiles_for_stype(args.stype)]
    per_lane = len(tile_ids)
    if args.tiles is None:
        args.tiles = per_lane * lpr
    tile_ids = (tile_ids * ((args.tiles + per_lane - 1) // per_lane))[:args.tiles]

    def lane_tiles_of(r):
        """(lane, tile) of every row rank r owns: its lanes one after the other; ids that repeat beyond them
        (only if --tiles exceeds the rank's lanes) get lanes of their own so that no two tiles share data."""
        first = args.first_lane + r * lpr
        return [(first + j if j < lpr else first + (j % lpr) + 8 * max(1, world * lpr // 8 + 1) * (j // lpr), t)
                for j, t in ((i // per_lane, t) for i, t in enumerate(tile_ids))]
    lane_tile = lane_tiles_of(rank)

    # One explicit stream for everything in a step (scan kernels, torch ops, the collective's
    # stream dependencies).  The scanner must NOT be left on its own stream here: torch's
    # zero_/all_reduce would then race with the scan.
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    sc = Scanner(local_rank)
    assert stream.cuda_stream != 0
    sc.set_stream(stream.cuda_stream)
    for opt in args.option:
        name, val = opt.split("=")
        sc.set_option(name, int(val))
    if args.no_early_exit:
        sc.set_option("early_exit", 0)
    sc.set_targets(centre, lvl_off, nbr)
    tb = TileBatch(sc, args.tiles, L, n_clusters)
    tb.fill_synthetic(spec, lane_tile, list(range(L)))
    setup_s = time.time() - t0

    # The job's counter block: one [world*tiles, ncnt] slab per step (= per lane batch), every
    # rank owning rows [rank*tiles, (rank+1)*tiles) of each slab.  As in the CLI
    # (count_well_duplicates.py of this package: one merge per run) the ranks' rows are merged
    # by ONE int64 all-reduce per job, i.e. per `--merge-every` steps (default: all K timed
    # steps, at most 256 slabs = 41 MB at 8 GPUs); `--merge-every 1` merges after every step.
    # Two job blocks alternate so that a merge (RCCL runs it on its own stream) overlaps the
    # scans of the next chunk.
    chunk = max(1, min(args.merge_every if args.merge_every > 0 else max(args.steps, 1), 256))
    jobs = [torch.zeros((chunk, world * args.tiles, ncnt), dtype=torch.int64, device="cuda") for _ in range(2)]
    pending = [None, None]
    turn = [0]
    # this rank's rows of step 0 of either block, and the bytes from one step's slab to the next
    slab_bytes = world * args.tiles * ncnt * 8
    row_ptr = [j.data_ptr() + rank * args.tiles * ncnt * 8 for j in jobs]

    def merge(buf, used):
        view = jobs[buf][:used]
        if rehearsal:
            host = view.cpu()
            dist.all_reduce(host)
            view.copy_(host)
        else:
            # RCCL int64 sum over xGMI; the ranks' rows are disjoint
            pending[buf] = dist.all_reduce(view, async_op=True)

    def finish_collectives():
        for i, w in enumerate(pending):
            if w is not None:
                w.wait()
                pending[i] = None

    def run(n_steps):
        """n_steps scans, merged chunk by chunk; returns the slab of the last step."""
        done, last = 0, None
        while done < n_steps:
            buf = turn[0]
            turn[0] ^= 1
            if pending[buf] is not None:
                pending[buf].wait()             # this block's previous merge (two chunks ago)
                pending[buf] = None
            n = min(chunk, n_steps - done)
            if use_dist:
                jobs[buf][:n].zero_()           # the other ranks' rows (a scan clears its own)
            for s_i in range(n):                # (addresses worked out ahead: the loop only launches)
                sc.scan_async(tb.tables, args.tiles, L, n_clusters, mode, k, row_ptr[buf] + s_i * slab_bytes)
            if use_dist:
                merge(buf, n)
            done += n
            last = jobs[buf][n - 1]
        finish_collectives()
        return last

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup)
    sc.scan_status()
    # HIP events around every n-th scan launch of the timed region, on the launch stream (a pair of event
    # records around EVERY launch costs the step 6 % - measured: 0.1355 against 0.1278 ms at 20 steps - so
    # at the driver's 20 steps every fourth launch is timed, at 200 every eighth: 5 and 25 launches), and
    # one more pair around the whole region on the same stream: the GPU-side time of all K launches
    # with their counter memsets, the cross-check of the per-launch figure
    sc.set_option("profile", int(os.environ.get("WD_BENCH_PROFILE_EVERY", 0)) or max(1, min(8, args.steps // 5)))
    sc.profile_reset()
    region = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    fence()
    t_start = time.perf_counter()
    region[0].record(stream)
    block = run(args.steps)
    region[1].record(stream)
    fence()
    elapsed = time.perf_counter() - t_start
    region_ms = region[0].elapsed_time(region[1]) / max(1, args.steps)
    kern_ms_total, launches = sc.profile_get()
    headline_kernel = sc.last_kernel()
    sc.set_option("profile", 0)
    sc.scan_status()
    elapsed = wdist.max_over_ranks(elapsed, world, device="cpu" if rehearsal else "cuda")   # slowest rank

    if block is None:                                           # --steps 0
        block = run(1)
        fence()
    my_rows = block[rank * args.tiles:(rank + 1) * args.tiles]
    counts = block.cpu().numpy()
    if rank == 0 and args.dump_block:
        np.save(args.dump_block, counts)
    compares_all = int(counts[:, 1:1 + levels].sum())          # sum of Wells over every rank
    mine = counts[rank * args.tiles:(rank + 1) * args.tiles]
    compares_rank = int(mine[:, 1:1 + levels].sum())
    valid_rank = int(mine[:, 0].sum())
    ms_per_step = elapsed / max(1, args.steps) * 1e3
    value = compares_all / (elapsed / max(1, args.steps))

    # ---- roofline of the dominant kernel: HIP events on the launch stream ----------------
    sc.set_option("profile", 1)
    if launches == 0:                                           # --steps 0
        sc.profile_reset()
        for _ in range(max(1, args.profile_steps)):
            sc.scan_async(tb.tables, args.tiles, L, n_clusters, mode, k, my_rows.data_ptr())
        kern_ms_total, launches = sc.profile_get()
    kern_ms = kern_ms_total / max(1, launches)
    # control for the Infinity Cache: the timed loop rescans ONE resident lane (0.77 GB of lines per
    # step out of 20.7 GB, the MALL holds 256 MiB), so some of its misses may never reach HBM.  The same
    # launches alternating between two resident lanes (41 GB) halve whatever the cache can keep from
    # one step to the next; if the kernel time does not move, the headline is an HBM figure.
    two_lanes = None
    if rank == 0 and world == 1 and args.profile_steps > 0 and not args.no_early_exit and args.tiles >= 8:
        tb2 = TileBatch(sc, args.tiles, L, n_clusters)
        tb2.fill_synthetic(spec, [(ln + 1, t) for ln, t in lane_tile], list(range(L)))
        scratch2 = torch.zeros((args.tiles, ncnt), dtype=torch.int64, device="cuda")
        for tables in (tb.tables, tb2.tables):
            sc.scan_async(tables, args.tiles, L, n_clusters, mode, k, scratch2.data_ptr())
        sc.profile_reset()
        t_a = time.perf_counter()
        for i_s in range(20):
            sc.scan_async((tb.tables, tb2.tables)[i_s & 1], args.tiles, L, n_clusters, mode, k, scratch2.data_ptr())
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t_a) / 20 * 1e3
        t_ms, t_n = sc.profile_get()
        sc.scan_status()
        tb2.free()
        two_lanes = {"kernel_ms": round(t_ms / max(1, t_n), 5), "ms_per_step": round(wall, 5), "launches_timed": t_n,
                     "resident_bytes": 2 * args.tiles * L * n_clusters,
                     "one_lane_kernel_ms": round(kern_ms, 5),
                     "ratio_to_one_lane": round(t_ms / max(1, t_n) / kern_ms, 4) if kern_ms > 0 else None,
                     "what": "20 launches alternating between two resident lanes (the timed region rescans one)"}
    # worst case for the lazy gather: nothing may die early (what low-diversity reads cost);
    # same counters, measured the same way, reported beside the headline for transparency
    worst = None
    if not args.no_early_exit and args.profile_steps > 0:
        sc.set_option("early_exit", 0)
        sc.scan_async(tb.tables, args.tiles, L, n_clusters, mode, k, my_rows.data_ptr())
        sc.profile_reset()
        for _ in range(3):
            sc.scan_async(tb.tables, args.tiles, L, n_clusters, mode, k, my_rows.data_ptr())
        w_ms, w_n = sc.profile_get()
        sc.set_option("early_exit", 1)
        worst = w_ms / max(1, w_n)
    # the reference's other compare modes on the same resident tiles (kernel time only):
    # Hamming <= 2 and the production default, Levenshtein <= 2
    other = {}
    if args.profile_steps > 0 and args.mode == "eq" and not args.no_early_exit:
        for name, m2, k2 in (("hamming_k2", MODE_HAMMING, 2), ("levenshtein_k2", MODE_LEVENSHTEIN, 2)):
            sc.scan_async(tb.tables, args.tiles, L, n_clusters, m2, k2, my_rows.data_ptr())
            sc.profile_reset()
            for _ in range(5):
                sc.scan_async(tb.tables, args.tiles, L, n_clusters, m2, k2, my_rows.data_ptr())
            o_ms, o_n = sc.profile_get()
            o_ms /= max(1, o_n)
            b_o = compares_rank * (L + 4) + valid_rank * (L + 5) + 8 * ncnt * args.tiles
            other[name] = with_traffic(
                {"kernel_ms": round(o_ms, 5), "compares_per_s": round(compares_rank / (o_ms * 1e-3), 1),
                 "alg_bytes_over_peak": round(b_o / (o_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                "%s_T%d_l%d_L%d" % ({"hamming_k2": "ham2", "levenshtein_k2": "lev2"}[name], T, levels, L), args.tiles, o_ms,
                sc.last_kernel())
        sc.scan_async(tb.tables, args.tiles, L, n_clusters, mode, k, my_rows.data_ptr())   # restore counters
        sc.scan_status()
        # the resident-layout option: the same tiles with their cycles interleaved by four
        # (what the loaders could write at no extra cost; include/welldup.h, wd_interleave4)
        if rank == 0 and world == 1 and args.interleaved_tiles > 0:
            n_il = min(args.interleaved_tiles, args.tiles)
            il = TileBatch(sc, n_il, L, n_clusters, interleave=4)
            il.fill_synthetic(spec, lane_tile[:n_il], list(range(L)))
            scratch = torch.zeros((n_il, ncnt), dtype=torch.int64, device="cuda")
            sc.set_option("well_stride", 4)
            sc.scan_async(il.tables, n_il, L, n_clusters, mode, k, scratch.data_ptr())
            sc.profile_reset()
            for _ in range(10):
                sc.scan_async(il.tables, n_il, L, n_clusters, mode, k, scratch.data_ptr())
            i_ms, i_n = sc.profile_get()
            il_kernel = sc.last_kernel()
            sc.set_option("well_stride", 1)
            sc.scan_status()
            torch.cuda.synchronize()
            same = bool((scratch.cpu().numpy() == mine[:n_il]).all())
            i_ms /= max(1, i_n)
            c_il = int(mine[:n_il, 1:1 + levels].sum())
            b_il = c_il * (L + 4) + int(mine[:n_il, 0].sum()) * (L + 5) + 8 * ncnt * n_il
            other["interleaved_by_4"] = with_traffic(
                {"tiles": n_il, "kernel_ms": round(i_ms, 5), "compares_per_s": round(c_il / (i_ms * 1e-3), 1),
                 "alg_bytes_over_peak": round(b_il / (i_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                 "same_counters_as_plane_layout": same},
                "il_T%d_l%d_L%d" % (T, levels, L), n_il, i_ms, il_kernel)
            # the reference's default metric in that layout
            sc.set_option("well_stride", 4)
            sc.scan_async(il.tables, n_il, L, n_clusters, MODE_LEVENSHTEIN, 2, scratch.data_ptr())
            sc.profile_reset()
            for _ in range(5):
                sc.scan_async(il.tables, n_il, L, n_clusters, MODE_LEVENSHTEIN, 2, scratch.data_ptr())
            l_ms, l_n = sc.profile_get()
            il_kernel = sc.last_kernel()
            sc.set_option("well_stride", 1)
            sc.scan_status()
            l_ms /= max(1, l_n)
            other["interleaved_by_4"]["levenshtein_k2"] = with_traffic(
                {"kernel_ms": round(l_ms, 5), "compares_per_s": round(c_il / (l_ms * 1e-3), 1),
                 "alg_bytes_over_peak": round(b_il / (l_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                "il_lev2_T%d_l%d_L%d" % (T, levels, L), n_il, l_ms, il_kernel)
            il.free()
            # what a run of the CLI keeps resident and scans (--layout auto: count_well_duplicates.py of this
            # package, resident_layout): the interleaved layout for the reference's default metric
            other["cli_default_path"] = {
                "layout": "interleaved_by_4 (--layout auto: sampled targets, -e <= 3 or --hamming, .bcl.gz runs)",
                "levenshtein_k2": dict(other["interleaved_by_4"]["levenshtein_k2"]),
                "planes_levenshtein_k2_kernel_ms": other.get("levenshtein_k2", {}).get("kernel_ms")}
        # the lazy gather's worst input: every read equal (amplicons, failed cycles - here all no-calls),
        # so that no neighbour ever dies early; 8 tiles, the headline's targets, both layouts
        if rank == 0 and world == 1 and args.interleaved_tiles > 0:
            n_ld = min(8, args.tiles)
            flat = synth.SynthSpec(seed=2, n_clusters=n_clusters, row=cols, nocall_per_64k=65536)
            scratch = torch.zeros((n_ld, ncnt), dtype=torch.int64, device="cuda")
            low = {"what": "%d tiles, every read equal: each of the %.0f neighbours of a target is a duplicate and is "
                           "read to its last cycle" % (n_ld, compares_rank / max(1, valid_rank)),
                   "ordinary_reads_us_per_tile": round(kern_ms / args.tiles * 1e3, 2)}
            for stride, key in ((1, "planes"), (4, "interleaved_by_4")):
                ld = TileBatch(sc, n_ld, L, n_clusters, interleave=stride)
                ld.fill_synthetic(flat, lane_tile[:n_ld], list(range(L)))
                sc.set_option("well_stride", stride)
                for name, m2, k2 in (("equality", MODE_EQ, 0), ("levenshtein_k2", MODE_LEVENSHTEIN, 2)):
                    sc.scan_async(ld.tables, n_ld, L, n_clusters, m2, k2, scratch.data_ptr())
                    sc.profile_reset()
                    for _ in range(5):
                        sc.scan_async(ld.tables, n_ld, L, n_clusters, m2, k2, scratch.data_ptr())
                    w_ms2, w_n2 = sc.profile_get()
                    low["%s_%s_us_per_tile" % (key, name)] = round(w_ms2 / max(1, w_n2) / n_ld * 1e3, 2)
                sc.set_option("well_stride", 1)
                sc.scan_status()
                ld.free()
            torch.cuda.synchronize()
            low["all_duplicates"] = bool((scratch.cpu().numpy()[:, 1 + levels:1 + 2 * levels]
                                          == scratch.cpu().numpy()[:, 1:1 + levels]).all())
            # ... and on a whole lane, as the headline is (8 tiles do not fill the chip), beside the time the
            # same loads take when nothing is left out on ordinary reads (the full gather: every neighbour's
            # every cycle is what the worst case has to read too)
            if args.tiles > n_ld:
                ld = TileBatch(sc, args.tiles, L, n_clusters)
                ld.fill_synthetic(flat, lane_tile[:args.tiles], list(range(L)))
                big = torch.zeros((args.tiles, ncnt), dtype=torch.int64, device="cuda")
                for name, m2, k2 in (("equality", MODE_EQ, 0), ("levenshtein_k2", MODE_LEVENSHTEIN, 2)):
                    sc.scan_async(ld.tables, args.tiles, L, n_clusters, m2, k2, big.data_ptr())
                    sc.profile_reset()
                    for _ in range(3):
                        sc.scan_async(ld.tables, args.tiles, L, n_clusters, m2, k2, big.data_ptr())
                    w_ms2, w_n2 = sc.profile_get()
                    low["lane_planes_%s_us_per_tile" % name] = round(w_ms2 / max(1, w_n2) / args.tiles * 1e3, 2)
                sc.scan_status()
                torch.cuda.synchronize()
                ld.free()
                low["lane_tiles"] = args.tiles
                low["lane_full_gather_on_ordinary_reads_us_per_tile"] = None if worst is None else round(worst / args.tiles * 1e3, 2)
            other["low_diversity_worst_case"] = low
    sc.set_option("profile", 0)
    # BASELINE configs[4] in small: every well of a tile is a centre (device-generated rings,
    # 3 levels), 150 bp, 2 % planted duplicates as in SURVEY.md 8d; its own context and planes
    if rank == 0 and world == 1 and args.dense_tiles > 0 and args.profile_steps > 0 and args.mode == "eq":
        other["dense_all_centres"] = dense_probe(local_rank, args.dense_tiles, rows, cols)
        if args.dense_tiles_large > args.dense_tiles:
            # the same chain over a whole lane (62 GB of planes resident): every kernel of it is a bigger launch, its
            # ramp and tail a smaller share - the per-tile figure of a real scan is this one
            other["dense_all_centres"]["at_%d_tiles" % args.dense_tiles_large] = dense_probe(
                local_rank, args.dense_tiles_large, rows, cols, modes=("levenshtein_k2", "equality"))
    if rank == 0 and world == 1 and args.novaseq_tiles > 0 and args.profile_steps > 0 and args.mode == "eq":
        other["novaseq_cfg4"] = novaseq_probe(local_rank, args.novaseq_tiles)
    b_alg = compares_rank * (L + 4) + valid_rank * (L + 5) + 8 * ncnt * args.tiles
    achieved = b_alg / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    # what THIS box's HBM gives a kernel that only reads (the boxes of a pool differ by several percent): the
    # resident lane's planes read once per pass by k_stream_read, nothing computed
    stream_read = None
    if rank == 0 and args.profile_steps > 0 and tb.plane_bytes >= (1 << 28):
        gbs = sc.stream_read_gbs(tb.d_planes, tb.plane_bytes - tb.plane_bytes % 16, passes=5)
        stream_read = {"gbs": round(gbs, 1), "bytes_per_pass": tb.plane_bytes - tb.plane_bytes % 16, "passes": 5,
                       "achieved_over_stream_read": round(achieved / gbs, 4) if gbs > 0 else None,
                       "what": "wd_stream_read_probe: the timed lane's resident planes read once per pass, 16 bytes per "
                               "lane and load, non-temporal, nothing computed - the rate a scan kernel's lines could "
                               "arrive at on this box (the guide's figure for a copy: %.0f GB/s)" % (ACHIEVABLE_HBM_FRAC * HBM_PEAK_GBS)}
    case = {"eq": "eq", "hamming": "ham%d" % k, "levenshtein": "lev%d" % k}[args.mode]
    if args.no_early_exit:
        case = "full"
    traffic, traffic_source = traffic_lookup("%s_T%d_l%d_L%d" % (case, T, levels, L), args.tiles, headline_kernel)
    roofline = {"bound": "hbm", "kernel": headline_kernel, "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                **traffic_fields(traffic, traffic_source, kern_ms),
                "build_id": __import__("well_duplicates_amd._lib", fromlist=["_lib"]).build_ids(),
                "achieved_is": "ALGORITHMIC bytes per launch (SURVEY.md 8d: 54 B per compare + 55 B per valid target "
                               "+ the counter rows) / the kernel's mean duration - the figure the contract asks for, not "
                               "a counter reading: a lazy kernel moves fewer bytes than the model charges, `traffic` / "
                               "`l2_miss_frac` beside it are the measured ones",
                "traffic_is": "L2-miss bytes per launch (2 x FETCH_SIZE + WRITE_SIZE of the PMC passes): an upper bound of "
                              "the HBM bytes, the Infinity Cache sits behind the L2",
                "two_lanes_alternating": two_lanes,
                "stream_read": stream_read,
                "algorithmic_bytes_per_launch": b_alg,
                "kernel_ms": round(kern_ms, 5), "launches_timed": launches,
                "region_ms_per_launch": round(region_ms, 5),
                "units_per_launch": compares_rank,
                "full_gather_kernel_ms": None if worst is None else round(worst, 5),
                "full_gather_alg_bytes_over_peak": None if not worst else round(b_alg / (worst * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

    # ---- CPU baseline: the oracle on a bounded sample, rank 0 at N = 1 only --------------
    cpu = cpu_py = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle
        n_cpu = max(1, min(args.cpu_tiles, args.tiles))
        usable = os.cpu_count() or 1
        try:                                                  # (the cgroup's quota, where there is one)
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if quota != "max":
                usable = min(usable, max(1, int(quota) // int(period)))
        except (OSError, ValueError):
            pass
        cores = max(1, min(32, usable, n_cpu))
        planes = [[tb.download_plane(i, c) for c in range(L)] for i in range(n_cpu)]
        filters = [tb.download_filter(i) for i in range(n_cpu)]
        reps, t_cpu, out = 0, 0.0, None
        t1 = time.perf_counter()
        while t_cpu < 3.0 and reps < 400:                # ~3 s wall x `cores` threads of CPU work
            out = oracle.count_tiles_mt(planes, filters, centre, lvl_off, nbr, mode, k, cores)
            reps += 1
            t_cpu = time.perf_counter() - t1
        cpu_compares = int(out[:, 1:1 + levels].sum())
        # the sample doubles as an end-to-end check of the timed path
        dev = mine[:n_cpu].copy()
        dev[:, 1 + 3 * levels:1 + 4 * levels] = np.cumsum(dev[:, 1 + 3 * levels:1 + 4 * levels], axis=1)
        dev[:, 1 + 4 * levels:] = np.cumsum(dev[:, 1 + 4 * levels:][:, ::-1], axis=1)[:, ::-1]
        if not (dev == out).all():
            raise SystemExit("bench: device counters differ from the CPU oracle on the sample tiles")
        cpu = {"value": round(cpu_compares * reps / t_cpu, 1), "unit": "compares/s", "cores": cores,
               "kind": "port",
               "sample": "%d of the %d tiles (%d compares), C oracle -O3, OpenMP over tiles, %d passes in %.1f s; "
                         "planes already decompressed in RAM (the reference also gunzips and builds Python strings)"
                         % (n_cpu, args.tiles, cpu_compares, reps, t_cpu),
               "parity_checked_tiles": n_cpu}
        # the faithful single-threaded Python restatement (reference structure: dict of
        # strings, per-base gather loop, triple compare loop) on a slice of one tile
        n_py = min(200, T)
        coords = [[[int(centre[t])]] + [nbr[lvl_off[t, l]:lvl_off[t, l + 1]].tolist() for l in range(levels)]
                  for t in range(n_py)]
        wells = [w for c in coords for ring in c for w in ring]
        pl_bytes = [p.tobytes() for p in planes[0]]
        t2 = time.perf_counter()
        seqs = oracle.py_get_seqs(pl_bytes, filters[0].tobytes(), wells)
        metric = {"eq": oracle.py_hamming, "hamming": oracle.py_hamming,
                  "levenshtein": oracle.py_levenshtein}[args.mode]
        stats = oracle.py_count_tile(coords, seqs, levels, metric, k)
        t_py = time.perf_counter() - t2
        py_compares = sum(ln for targ in stats for _, ln in targ)
        cpu_py = {"value": round(py_compares / t_py, 1), "unit": "compares/s", "cores": 1, "kind": "port",
                  "sample": "first %d targets of 1 tile (%d compares) in %.2f s, pure-Python restatement of "
                            "get_seqs + the compare loop, planes already gunzipped" % (n_py, py_compares, t_py)}

    # ---- --check-tiles: sampled rows of the MERGED block (every rank's) against the oracle ----
    checked = None
    if rank == 0 and args.check_tiles > 0:
        from oracle import oracle
        rng = np.random.default_rng(5)
        total_rows = world * args.tiles
        pick = sorted(rng.choice(total_rows, size=min(args.check_tiles, total_rows), replace=False).tolist())
        wells_u = np.unique(np.concatenate([centre, nbr]).astype(np.int64))
        c2 = np.searchsorted(wells_u, centre).astype(np.int32)
        n2 = np.searchsorted(wells_u, nbr).astype(np.int32)
        planes_h, filters_h = [], []
        for row in pick:                     # (only the wells the targets touch: the oracle reads no others)
            ln, t = lane_tiles_of(row // args.tiles)[row % args.tiles]
            planes_h.append([synth.plane_bytes(spec, ln, t, c, wells_u) for c in range(L)])
            filters_h.append(synth.filter_bytes(spec, ln, t, wells_u))
        out = oracle.count_tiles_mt(planes_h, filters_h, c2, lvl_off, n2, mode, k, min(16, len(pick)))
        dev = counts[pick].copy()
        dev[:, 1 + 3 * levels:1 + 4 * levels] = np.cumsum(dev[:, 1 + 3 * levels:1 + 4 * levels], axis=1)
        dev[:, 1 + 4 * levels:] = np.cumsum(dev[:, 1 + 4 * levels:][:, ::-1], axis=1)[:, ::-1]
        if not (dev == out).all():
            raise SystemExit("bench: merged block differs from the CPU oracle on sampled tiles %s" % pick)
        checked = {"rows": pick, "ranks_covered": sorted({r // args.tiles for r in pick}), "oracle": "count_tiles_mt"}

    e2e = None
    if rank == 0 and world == 1 and args.e2e_tiles > 0 and args.profile_steps > 0 and args.mode == "eq":
        tb.free()                                             # (the e2e run brings its own tiles)
        tb = None
        e2e = e2e_probe(local_rank, args.e2e_tiles, rows, cols, centre, lvl_off, nbr)
    if rank == 0:
        per_lane = len(workload.tiles_for_stype(args.stype))
        named = {("hiseq_x", 96, 1): "BASELINE configs[1]", ("hiseq_4000", 112, 1): "BASELINE configs[2] layout"}.get(
            (args.stype, args.tiles, lpr), "custom layout")
        if args.stype == "hiseq_4000" and world * lpr == 8 and args.tiles == 112 * lpr:
            named = "BASELINE configs[2]: 8 lanes x 112 tiles" + ("" if lpr == 1 else ", %d lanes per rank" % lpr)
        line = {
            "metric": "target x neighbour seq-compares/sec (whole node)",
            "value": round(value, 1), "unit": "compares/s",
            "n_gpus": world, "rccl_world": None if ranks_seen is None else ranks_seen["rccl_world"],
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%d lane(s) x %d tiles (%s) x %d targets x %d levels, %d bp; %s per GPU (%s)"
                                   % (world * lpr, args.tiles // lpr, args.stype, T, levels, L,
                                      "one lane" if lpr == 1 else "%d lanes" % lpr, named),
                       "lanes_per_rank": lpr, "resident_bytes_per_rank": args.tiles * (L + 1) * n_clusters,
                       "checked_against_oracle": checked,
                       "mode": args.mode, "k": k, "early_exit": not args.no_early_exit,
                       "clusters_per_tile": n_clusters, "compares_per_step": compares_all,
                       "valid_targets_per_rank": valid_rank, "parallelism": "tiles sharded, %d rank(s)" % world,
                       "merge": None if not use_dist else "one int64 all-reduce per %d step(s)" % chunk,
                       "process_group": ranks_seen,
                       "setup_s": round(setup_s, 1)},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "cpu_baseline_python": cpu_py,
            "e2e": e2e,
            "other_modes": other,
        }
        print(json.dumps(line))
    if tb is not None:
        tb.free()
    sc.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
