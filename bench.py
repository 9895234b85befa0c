#!/usr/bin/env python3
"""bench.py - the headline benchmark: target x neighbour sequence compares per second.

Workload (BASELINE.json configs[1], one per rank = weak scaling): one lane of a HiSeq-X style
flowcell = 96 tiles x 2500 targets (prepare_cluster_indexes semantics, seed 13) x 5 levels,
50 bp, HiSeq-4000 tile geometry (2743 x 1571 = 4 309 253 clusters/tile), synthetic BCL planes
and filters generated on the device from the counter-based spec (well_duplicates_amd/synth.py:
0.5 % no-calls, 70 % filter pass, 2 % planted near-duplicates).  Rank r scans lane r+1.

A "step" is one pass of the scan path over the rank's 96 resident tiles (all kernels of
wd_scan_async) into that step's rows of the job's counter block.  For N > 1 the ranks' rows are
merged inside the timed region by one int64 all-reduce of the [steps, lanes*tiles, 1+5*levels]
block per job (ONE collective per job, as in the CLI; `--merge-every 1` merges
after every step instead).  Inputs are resident in HBM before the timed region.  One unit of work is one
performed compare = one unit of the reference report's `Wells` column
(count_well_duplicates.py:93), read back from the device's own counters.

Extra objects on the JSON line:
  roofline      HBM roofline of the dominant kernel (k_scan): algorithmic bytes
                B = C*(L+4) + Tv*(L+5) + 8*(1+5*levels)*tiles per launch over the
                kernel's mean duration, measured with HIP events on the launch stream around
                every 8th launch of the timed region itself.
  cpu_baseline  the C oracle (oracle/welldup_oracle.c, -O3, OpenMP over tiles) on a bounded
                sample of the same tiles, on this box's host cores.
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--tiles", type=int, default=None,
                    help="tiles per rank (one lane); default: all tiles of --stype (96 / 112)")
    ap.add_argument("--stype", default="hiseq_x", choices=["hiseq_x", "hiseq_4000"],
                    help="lane layout: hiseq_x = 96 tiles (BASELINE configs[1]), hiseq_4000 = 112 (configs[2])")
    ap.add_argument("--targets", type=int, default=2500)
    ap.add_argument("--levels", type=int, default=5)
    ap.add_argument("--bases", type=int, default=50)
    ap.add_argument("--mode", default="eq", choices=["eq", "hamming", "levenshtein"])
    ap.add_argument("-k", type=int, default=0, help="distance threshold (hamming / levenshtein)")
    ap.add_argument("--no-early-exit", action="store_true",
                    help="full gather: read all L bytes of every neighbour")
    ap.add_argument("--cpu-tiles", type=int, default=8, help="tiles in the CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=20)
    ap.add_argument("--interleaved-tiles", type=int, default=96,
                    help="tiles of the interleaved-layout measurement under other_modes; 0 = skip")
    ap.add_argument("--dense-tiles", type=int, default=4,
                    help="tiles of the all-centres probe (BASELINE configs[4] shape: every well a centre, "
                         "3 levels, 150 bp) reported under other_modes; 0 = skip")
    ap.add_argument("--option", action="append", default=[], help="name=value scanner option")
    ap.add_argument("--merge-every", type=int, default=0,
                    help="steps merged by one all-reduce (N > 1); 0 = one merge per job (all timed steps)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1: nccl = RCCL over xGMI (default); gloo = "
                         "rehearsal of the multi-rank logic with several ranks on ONE GPU")
    return ap.parse_args()


def dense_probe(device, n_tiles, rows, cols, levels=3, bases=150):
    """Kernel time of the dense path (scan_dense.inc) on n_tiles full-size tiles."""
    import numpy as np
    from well_duplicates_amd import synth
    from well_duplicates_amd.scanner import Scanner, TileBatch, MODE_EQ, MODE_LEVENSHTEIN
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
        sc.scan_async(tb.tables, n_tiles, bases, n, MODE_EQ, 0, out)      # builds the tables
        sc.set_option("profile", 1)
        # the reference's default metric first (Levenshtein <= 2), then equality for the counters
        sc.scan_async(tb.tables, n_tiles, bases, n, MODE_LEVENSHTEIN, 2, out)
        sc.profile_reset()
        for _ in range(3):
            sc.scan_async(tb.tables, n_tiles, bases, n, MODE_LEVENSHTEIN, 2, out)
        lev_ms, lev_n = sc.profile_get()
        lev_ms /= max(1, lev_n)
        sc.profile_reset()
        for _ in range(3):
            sc.scan_async(tb.tables, n_tiles, bases, n, MODE_EQ, 0, out)
        ms, launches = sc.profile_get()
        ms /= max(1, launches)
        blk = sc.d2h(out, n_tiles * ncnt * 8, np.int64).reshape(n_tiles, ncnt)
        compares = int(blk[:, 1:1 + levels].sum())
        b_dense = n_tiles * (n * bases + 4 * n * (1 + P / T) + n)             # SURVEY.md 8d, dense form
        res = {"workload": "%d tiles x %d centres x %.1f neighbours, %d levels, %d bp, 2 %% planted"
                           % (n_tiles, T, P / T, levels, bases),
               "kernel_ms": round(ms, 4), "ms_per_tile": round(ms / n_tiles, 4),
               "compares_per_s": round(compares / (ms * 1e-3), 1),
               "duplicates_found": int(blk[:, 1 + levels:1 + 2 * levels].sum()),
               "algorithmic_bytes": int(b_dense), "frac_of_hbm_peak": round(b_dense / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
               "levenshtein_k2_kernel_ms": round(lev_ms, 4),
               "levenshtein_k2_compares_per_s": round(compares / (lev_ms * 1e-3), 1),
               "ring_generator_s": round(gen_s, 3)}
        tb.free()
        return res
    finally:
        sc.close()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d"
                     % (args.gpus, args.gpus))
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
    lane = rank + 1
    tile_ids = [int(t) for t in workload.tiles_for_stype(args.stype)]
    if args.tiles is None:
        args.tiles = len(tile_ids)
    per_lane = len(tile_ids)
    tile_ids = (tile_ids * ((args.tiles + per_lane - 1) // per_lane))[:args.tiles]
    # repeated ids (only if --tiles exceeds a lane) get distinct lanes so no two tiles share data
    lane_tile = [(lane + 8 * (i // per_lane), t) for i, t in enumerate(tile_ids)]

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
            for s_i in range(n):
                rows = jobs[buf][s_i, rank * args.tiles:(rank + 1) * args.tiles]
                sc.scan_async(tb.tables, args.tiles, L, n_clusters, mode, k, rows.data_ptr())
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
    # HIP events around every 8th scan launch of the timed region, on the launch stream: the
    # roofline's kernel time comes from launches that `value` is made of (every launch would cost
    # the step 3-5 % in event records)
    sc.set_option("profile", 8)
    sc.profile_reset()
    fence()
    t_start = time.perf_counter()
    block = run(args.steps)
    fence()
    elapsed = time.perf_counter() - t_start
    kern_ms_total, launches = sc.profile_get()
    sc.set_option("profile", 0)
    sc.scan_status()
    elapsed = wdist.max_over_ranks(elapsed, world, device="cpu" if rehearsal else "cuda")   # slowest rank

    if block is None:                                           # --steps 0
        block = run(1)
        fence()
    my_rows = block[rank * args.tiles:(rank + 1) * args.tiles]
    counts = block.cpu().numpy()
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
            other[name] = {"kernel_ms": round(o_ms / max(1, o_n), 5),
                           "compares_per_s": round(compares_rank / (o_ms / max(1, o_n) * 1e-3), 1)}
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
            sc.set_option("well_stride", 1)
            sc.scan_status()
            torch.cuda.synchronize()
            same = bool((scratch.cpu().numpy() == mine[:n_il]).all())
            i_ms /= max(1, i_n)
            c_il = int(mine[:n_il, 1:1 + levels].sum())
            b_il = c_il * (L + 4) + int(mine[:n_il, 0].sum()) * (L + 5) + 8 * ncnt * n_il
            other["interleaved_by_4"] = {
                "tiles": n_il, "kernel_ms": round(i_ms, 5), "compares_per_s": round(c_il / (i_ms * 1e-3), 1),
                "frac_of_hbm_peak": round(b_il / (i_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "same_counters_as_plane_layout": same}
            il.free()
    sc.set_option("profile", 0)
    # BASELINE configs[4] in small: every well of a tile is a centre (device-generated rings,
    # 3 levels), 150 bp, 2 % planted duplicates as in SURVEY.md 8d; its own context and planes
    if rank == 0 and world == 1 and args.dense_tiles > 0 and args.profile_steps > 0 and args.mode == "eq":
        other["dense_all_centres"] = dense_probe(local_rank, args.dense_tiles, rows, cols)
    b_alg = compares_rank * (L + 4) + valid_rank * (L + 5) + 8 * ncnt * args.tiles
    achieved = b_alg / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(REPO, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            key = "%s_k%d_t%d_T%d_l%d_L%d_%s" % (args.mode, k, args.tiles, T, levels, L,
                                                "full" if args.no_early_exit else "early")
            traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    kernel_name = "k_scan_q" if (args.mode != "levenshtein" or k < 2) and not args.no_early_exit else "k_scan"
    roofline = {"bound": "hbm", "kernel": kernel_name, "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic, "algorithmic_bytes_per_launch": b_alg,
                "kernel_ms": round(kern_ms, 5), "launches_timed": launches,
                "units_per_launch": compares_rank,
                "full_gather_kernel_ms": None if worst is None else round(worst, 5),
                "full_gather_frac": None if not worst else round(b_alg / (worst * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

    # ---- CPU baseline: the oracle on a bounded sample, rank 0 at N = 1 only --------------
    cpu = cpu_py = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle
        n_cpu = max(1, min(args.cpu_tiles, args.tiles))
        cores = max(1, min(16, os.cpu_count() or 1, n_cpu))
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

    if rank == 0:
        line = {
            "metric": "target x neighbour seq-compares/sec (whole node)",
            "value": round(value, 1), "unit": "compares/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "1 lane x %d tiles x %d targets x %d levels, %d bp per GPU "
                                   "(BASELINE configs[1]; N GPUs = N lanes)" % (args.tiles, T, levels, L),
                       "mode": args.mode, "k": k, "early_exit": not args.no_early_exit,
                       "clusters_per_tile": n_clusters, "compares_per_step": compares_all,
                       "valid_targets_per_rank": valid_rank, "parallelism": "tiles sharded, %d rank(s)" % world,
                       "merge": None if not use_dist else "one int64 all-reduce per %d step(s)" % chunk,
                       "setup_s": round(setup_s, 1)},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "cpu_baseline_python": cpu_py,
            "other_modes": other,
        }
        print(json.dumps(line))
    tb.free()
    sc.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
