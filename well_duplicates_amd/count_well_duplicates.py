#!/usr/bin/env python3
"""count_well_duplicates on an MI355X: same CLI, targets-file format and report as the
reference's count_well_duplicates.py, with the per-tile gather + compare + tally
(count_well_duplicates.py:212-265, :63-106) running in libwelldup.so.

Flow per lane (the reference's unit of output, :207-269):
  read the scanned cycles' planes + filter of every tile (threads: gunzip releases the GIL)
  -> upload into an HBM-resident TileBatch -> one wd_count_tiles call for the whole batch
  -> per-tile integer blocks -> report.write_report (text identical to output_writer).
Unless -q is given the stderr log is reproduced too, including the three lines the
reference prints per duplicate (:258-262), from the device's hit list.

Multi-GPU: started under torchrun (one process per GPU) the flat list of (lane, tile) items is
block-partitioned over the ranks, each rank scans its share on its own GPU, ONE int64
all-reduce per run merges the per-tile counter rows (well_duplicates_amd/dist.py) and rank 0
prints the lanes in order - identical to the single-GPU output.

Differences from the reference, all deliberate:
  * a lane with valid targets but no duplicate prints 0.00 % instead of dying with
    ZeroDivisionError (F5); --strict restores the exception;
  * `-t` with a pattern that matches nothing raises the AssertionError the reference
    intends (its own message formatting raises NameError first);
  * extra flags: --device, --dist-backend, --tile-batch, --threads, --strict, -o/--output,
    --all-wells, --slocs, --layout, --serial-ingest;
  * the resident layout is chosen per run (--layout auto): sampled scans the interleaved-by-four layout
    serves (the reference's default -e 2 among them) keep their cycles interleaved, everything else planes.
"""
from __future__ import annotations

import os
import sys
from argparse import ArgumentDefaultsHelpFormatter, ArgumentParser
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import bcl as bcl_direct_reader
from . import report, workload
from .report import LENGTH, TALLY, output_writer  # noqa: F401  (reference module surface)
from .scanner import INVALID_TARGET, Scanner, TileBatch, compare_mode
from .targets import load_targets, load_targets_csr

__VERSION__ = 0.3        # report format version of the reference this mirrors (:4)

HISEQ_4000 = workload.HISEQ_4000
HISEQ_X = workload.HISEQ_X
SEQUENCE = bcl_direct_reader.SEQUENCE
QUAL_FLAG = bcl_direct_reader.QUAL_FLAG


def parse_args(argv=None):
    """Same options as the reference (count_well_duplicates.py:272-317) plus device knobs."""
    description = """Assess well duplicates in a run without mapping. Reads within level l of
    selected reads from the coordinate file are compared with the centre read (edit distance,
    or Hamming distance with --hamming) on an AMD MI355X."""
    p = ArgumentParser(description=description, formatter_class=ArgumentDefaultsHelpFormatter)
    p.add_argument("-f", "--coord_file", dest="coord_file", required=False,
                   help="The file containing the random sample per tile (required unless --all-wells).")
    p.add_argument("-e", "--edit_distance", dest="edit_distance", type=int, default=2,
                   help="max edit distance between two reads to count as duplicate")
    p.add_argument("-n", "--sample_size", dest="sample_size", type=int, default=2500,
                   help="number of reads to be tested for well duplicates")
    p.add_argument("-l", "--level", dest="level", type=int, default=3,
                   help="levels around central spot to test")
    p.add_argument("-s", "--stype", dest="stype", required=True,
                   help="Sequencer model. Can be {} or {} or else the highest tile number in which "
                        "case the tile/swath configuration will be inferred.".format(HISEQ_4000, HISEQ_X))
    p.add_argument("-r", "--run", dest="run", required=True,
                   help="path to base of run, i.e /ifs/seqdata/150715_K00169_0016_BH3FGFBBXX")
    p.add_argument("-t", "--tile", dest="tile_id", type=str,
                   help="comma-separated list of specific tiles on a lane to analyse; each item "
                        "is a regex, so 1... is the top surface only.")
    p.add_argument("-i", "--lane", dest="lane", type=str,
                   help="comma-separated list of specific lanes to analyse, 1-8")
    p.add_argument("-x", "--start", dest="start", type=int, default=50,
                   help="Starting cycle/base position for the slice of read to be examined")
    p.add_argument("-y", "--end", dest="end", type=int, default=100,
                   help="Final cycle/base position for the slice of read to be examined")
    p.add_argument("--cycles",
                   help="Cycles/bases to scan as a list of ranges, eg. 10-50,100-120. Overrides -x/-y.")
    p.add_argument("--hamming", action="store_true",
                   help="Compare sequences using the Hamming distance rather than the "
                        "Levenshtein edit distance.")
    p.add_argument("-S", "--summary-only", action="store_true",
                   help="Only print the summary per lane, not for every tile")
    p.add_argument("-q", "--quiet", action="store_true", help="No log output")
    p.add_argument("--version", action="version", version=str(__VERSION__))
    p.add_argument("--device", type=int, default=None,
                   help="GPU to run on (default: LOCAL_RANK under torchrun, else 0)")
    p.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo", "wd"],
                   help="the one collective under torchrun: nccl = RCCL through torch.distributed; "
                        "wd = RCCL through libwelldup's own binding (wd_allreduce_counts; torch only carries "
                        "the unique id); gloo = CPU, for rehearsals")
    p.add_argument("--tile-batch", type=int, default=0,
                   help="tiles loaded, kept resident in HBM and scanned together (default 0: as many as make "
                        "about 512 files, what one launch of the GPU decoder holds at once)")
    p.add_argument("--threads", type=int, default=default_threads(),
                   help="reader threads (gunzip with --host-inflate)")
    p.add_argument("-o", "--output", default=None,
                   help="write the report to this file instead of stdout")
    p.add_argument("--all-wells", action="store_true",
                   help="every well of a tile is a centre (no sampling, no targets file): the rings of "
                        "-l levels are generated on the GPU from the run's s.locs with the rules of "
                        "prepare_cluster_indexes.py; the per-duplicate log is not written in this mode")
    p.add_argument("--slocs", default=None,
                   help="s.locs file for --all-wells (default: <run>/Data/Intensities/s.locs)")
    p.add_argument("--layout", default="auto", choices=["auto", "planes", "interleaved"],
                   help="how the scanned cycles sit in GPU memory: planes = one plane per cycle, as in the "
                        ".bcl.gz files; interleaved = the four cycles of a group side by side per well (the "
                        "loaders write it at no extra cost; the scan of sampled targets then touches half the "
                        "cache lines).  interleaved: -e <= 3 (any -e with --hamming up to 254), not "
                        "with --all-wells.  auto (default) = interleaved wherever that holds, else planes")
    p.add_argument("--host-inflate", action="store_true",
                   help="gunzip the .bcl.gz files on the host threads (the default inflates them on the GPU: "
                        "the threads only read the compressed files, one wave per file decodes it; files the "
                        "GPU decoder declines are gunzipped on the host either way)")
    p.add_argument("--serial-ingest", action="store_true",
                   help="load a batch of tiles only after the previous one has been scanned (for measuring "
                        "what the double-buffered ingest gains)")
    p.add_argument("--strict", action="store_true",
                   help="reproduce the reference's ZeroDivisionError on a lane without duplicates")
    args = p.parse_args(argv)
    if not args.coord_file and not args.all_wells:
        p.error("the following arguments are required: -f/--coord_file (or --all-wells)")
    if args.layout == "interleaved" and (args.all_wells or (args.edit_distance > 3 and not args.hamming)):
        p.error("--layout interleaved needs sampled targets (-f) and, for the edit distance, -e <= 3")
    return args


_T0 = [0.0]


def default_threads() -> int:
    """Reader threads (they read files into pinned memory; with --host-inflate they also gunzip):
    at most 32, and under torchrun this rank's share of the host's CPUs - eight ranks of one node
    read through one page cache and must not start 8 x 32 threads."""
    cpus = os.cpu_count() or 1
    try:
        cpus = len(os.sched_getaffinity(0)) or cpus
    except (AttributeError, OSError):
        pass
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1))
    return max(1, min(32, cpus // local_world))


def resident_layout(args, mode, k, csr, reader, lanes, tiles, cycle_list) -> int:
    """--layout -> the well stride of the batches (1 = a plane per cycle, 4 = interleaved by four).  `auto`
    takes the interleaved layout wherever the kernels that read it serve the run: sampled targets of at most
    508 neighbour slots, equality / Hamming <= 254 / Levenshtein <= 3 (the reference's default is 2:
    count_well_duplicates.py:282-283); .bcl.gz files and the .cbcl blocks of a NovaSeq run are both written
    straight into it by the loaders."""
    from .scanner import MODE_EQ, MODE_HAMMING
    if args.layout != "auto":
        return 4 if args.layout == "interleaved" else 1
    if args.all_wells or csr is None:
        return 1
    served = mode == MODE_EQ or (mode == MODE_HAMMING and k <= 254) or (mode not in (MODE_EQ, MODE_HAMMING) and k <= 3)
    lvl_off = csr[1]
    slots = int((lvl_off[:, -1] - lvl_off[:, 0]).max()) if lvl_off.shape[0] else 0
    if not served or slots > 508 or not cycle_list or lvl_off.shape[0] >= 65536:      # (65536 targets: the dense path, on planes)
        return 1
    return 4


def _lap(what: str):
    """WD_CLI_TIMING=1: where the wall clock of a run goes (stderr)."""
    if os.environ.get("WD_CLI_TIMING"):
        import time
        now = time.perf_counter()
        print("[wd timing] %-28s %7.1f ms" % (what, (now - _T0[0]) * 1e3), file=sys.stderr)
        _T0[0] = now


def _decode(seq_bytes: np.ndarray) -> str:
    """BCL bytes of one well over the scanned cycles -> the reference's string."""
    lut = np.frombuffer(b"NACGT", dtype="S1")
    return lut[np.where(seq_bytes == 0, 0, (seq_bytes & 3) + 1)].tobytes().decode()


def _decode_rows(rows: np.ndarray):
    """[n, L] BCL bytes -> n strings."""
    lut = np.frombuffer(b"NACGT", dtype="S1")
    codes = lut[np.where(rows == 0, 0, (rows & 3) + 1)]
    return [r.tobytes().decode() for r in codes]


class _Loading:
    """One batch of tiles on its way into HBM: the TileBatch and the loader pool's futures."""

    def __init__(self, chunk, handles, tb, futures, error=None):
        self.chunk, self.handles, self.tb, self.futures = chunk, handles, tb, futures
        self.error = error                  # what submitting the batch raised (a missing lane directory, ...)

    def wait(self):
        if self.error is not None:
            raise self.error
        for f in self.futures:
            f.result()                      # re-raises the loader's exception (FileNotFoundError, ...)


def scan_lanes(sc: Scanner, reader, lane_tiles, cycle_list, mode, k, csr, wells, tile_batch,
               threads, want_log, overlap=True, interleave=1, gpu_inflate=True, lane_done=None, into=None):
    """lane_tiles: [(lane, [tiles])] in the order they are reported -> ({(lane, tile): TileCounts},
    {(lane, tile): [log lines]}); `lane_done(lane)` is called when a lane's last tile has been scanned.

    Pipelined: while the GPU scans batch n (and its report rows and log lines are put together),
    batch n + 1 is being inflated and batch n + 2 read and copied, each into a TileBatch of its own
    (the ctypes calls release the GIL).  A batch holds tiles of one lane, but the pipeline runs on from
    one lane into the next: the first batches of lane n + 1 are on their way while the last ones of lane
    n are decoded and scanned (a lane at a time, every lane paid for the fill and the drain of its own
    pipeline: a tenth of a second of a 0.43 s lane).  Whatever goes wrong, every loader thread has
    finished before a TileBatch is freed or the exception leaves this function: the threads
    write through the scanner's copy streams into the batches' planes.
    """
    centre, lvl_off, nbr = csr
    levels = lvl_off.shape[1] - 1
    counts, logs = (into["counts"], into["logs"]) if into else ({}, {})
    batches = []                            # (lane, [tiles]), never across a lane's end: an error stays its lane's (start_ahead)
    for lane, tiles in lane_tiles:
        if not tiles:
            continue
        tb_n = tile_batch
        if tb_n <= 0:
            # as many files per batch as one launch of the GPU decoder holds at once - 512, or 768 of files
            # that expand less than 1.75-fold (its small-window form; base calls with binned qualities do:
            # the first file's size tells) - in batches of equal size (a short last batch would cost a full
            # round of the decoder all the same)
            room = 512
            try:
                first = reader.get_tile(lane, tiles[0])
                if os.path.getsize(first.plane_path(cycle_list[0])) * 7 >= (first.num_clusters + 4) * 4:
                    room = 700
            except (OSError, IndexError, RuntimeError, AssertionError):
                pass                            # (whatever is wrong with the first tile is reported when it is loaded)
            per = max(1, room // max(1, len(cycle_list) + 1))
            n_batches = max(1, -(-len(tiles) // per))
            tb_n = max(1, -(-len(tiles) // n_batches))
        batches += [(lane, tiles[b0:b0 + tb_n]) for b0 in range(0, len(tiles), tb_n)]
    last_batch_of = {lane: bi for bi, (lane, _) in enumerate(batches)}
    pool = ThreadPoolExecutor(max_workers=max(1, threads))
    live = []                               # TileBatches not yet freed
    spare = []                              # finished ones whose buffers the next batch takes over

    def start(batch):
        """Submit every load of a batch; returns at once."""
        lane, chunk = batch
        handles = [reader.get_tile(lane, t) for t in chunk]
        n_clusters = handles[0].num_clusters
        for h in handles:
            # one targets file, hence one geometry, per flowcell (README.md:14)
            if h.num_clusters != n_clusters:
                raise RuntimeError("tiles of one batch differ in cluster count")
        if wells.size and (wells[-1] >= n_clusters or wells[0] < 0):
            raise IndexError("Requested cluster %i is out of range.  Highest on this "
                             "tile is %i." % (int(wells[-1]), n_clusters - 1))
        # (a finished batch's buffers are taken over: freeing device memory would wait for the
        # decoder's kernels of the batches behind)
        tb = TileBatch(sc, len(chunk), len(cycle_list), n_clusters, interleave=interleave,
                       reuse=spare.pop() if spare else None)
        live.append(tb)
        _lap("  batch of %d tiles: buffers" % len(chunk))
        # ingest: every (tile, cycle) file is gunzipped into pinned memory and copied to the GPU
        # by libwelldup (wd_load_bcl_gz).  Runs without .bcl.gz files are NovaSeq runs: the
        # tile's block of the lane/surface .cbcl is gunzipped on the host and expanded on the
        # GPU (wd_load_cbcl_tile), which needs the tile's filter first.
        jobs = [(i, c) for i in range(len(handles)) for c in range(len(cycle_list))]
        batch = None                        # which files the GPU decoder gets as one batch
        if gpu_inflate and jobs:
            if os.path.exists(handles[0].plane_path(cycle_list[0])):
                batch = "bcl.gz"
            elif os.path.exists(handles[0].cbcl_path(cycle_list[0])):
                batch = "cbcl"
        # (in a batch the filters travel with the planes, below)
        filt = [] if batch else [pool.submit(sc.load_filter, h.filter_file, tb.filter_ptr(i), n_clusters)
                                 for i, h in enumerate(handles)]

        def load(i, c):
            try:
                sc.load_bcl_gz(handles[i].plane_path(cycle_list[c]), tb.plane_ptr(i, c), n_clusters, interleave)
            except FileNotFoundError:       # only a missing file: a corrupt one is reported as such
                if filt:
                    filt[i].result()
                sc.load_cbcl_tile(handles[i].cbcl_path(cycle_list[c]), int(handles[i].tile),
                                  tb.filter_ptr(i), n_clusters, tb.plane_ptr(i, c), interleave)
        if batch == "cbcl":
            # NovaSeq: the filters first (the expansion of a tile's blocks needs its filter in HBM),
            # then every (tile, cycle) block of the batch through one launch of the GPU decoder
            def load_all():
                sc.load_bcl_gz_batch([], [], n_clusters, threads=max(1, threads),
                                     filters=[(h.filter_file, tb.filter_ptr(i)) for i, h in enumerate(handles)])
                sc.load_cbcl_batch([(handles[i].cbcl_path(cycle_list[c]), int(handles[i].tile), tb.filter_ptr(i),
                                     tb.plane_ptr(i, c)) for i, c in jobs], n_clusters, threads=max(1, threads),
                                   well_stride=interleave)
            planes = [pool.submit(load_all)]
        elif batch:
            # the whole batch - planes and filters - goes through one call: the library's threads read
            # the files, the GPU inflates the .bcl.gz ones (wd_load_tile_files_batch)
            def load_all():
                missing = sc.load_bcl_gz_batch([handles[i].plane_path(cycle_list[c]) for i, c in jobs],
                                               [tb.plane_ptr(i, c) for i, c in jobs], n_clusters,
                                               threads=max(1, threads), missing_ok=True, well_stride=interleave,
                                               filters=[(h.filter_file, tb.filter_ptr(i)) for i, h in enumerate(handles)],
                                               tile_of=[i for i, _ in jobs] + list(range(len(handles))))
                for j in missing:           # (a run is .bcl.gz or .cbcl, never both: this loop is for the odd file)
                    load(*jobs[j])
            planes = [pool.submit(load_all)]
        else:
            planes = [pool.submit(load, i, c) for i, c in jobs]
        return _Loading((lane, chunk), handles, tb, filt + planes)

    def start_ahead(batch):
        """start(), for a batch that is not yet the current one: whatever submitting it raises (a lane
        directory that does not exist, no .filter file, a tile of another size, an index beyond the tile)
        is kept and raised when the batch's turn comes - the reference reports lane n before it touches
        lane n + 1 (count_well_duplicates.py:207-226, :269), so an error of a later lane must not cost
        an earlier lane its report."""
        try:
            return start(batch)
        except Exception as e:              # noqa: BLE001 - re-raised by _Loading.wait()
            return _Loading(batch, None, None, [], error=e)

    def release(tb, keep=False):
        live.remove(tb)
        if keep:
            spare.append(tb)
        else:
            tb.free()

    try:
        # three batches on their way at any time: one's files are read and copied while the one before
        # is still being inflated (the GPU decoder's time per batch does not shrink with the batch)
        # and the one before that is scanned and reported - the library serves the batch calls in
        # the order they were made
        depth = 3 if overlap else 1
        ahead = []
        for b in batches[:depth]:
            ahead.append(start_ahead(b))
        for bi in range(len(batches)):
            cur = ahead.pop(0)
            cur.wait()
            _lap("batch %d: planes in HBM" % bi)
            (lane, chunk), tb = cur.chunk, cur.tb
            n_clusters = tb.N
            if want_log:
                sc.hitlog_enable(max(1024, int(nbr.size) * len(chunk)))
            blocks, _ = tb.count(mode, k)
            _lap("batch %d: scanned" % bi)
            hits, seq_bytes, seq_wells = None, {}, {}
            if want_log:
                hits, total = sc.hitlog_fetch(max(1024, int(nbr.size) * len(chunk)))
                sc.hitlog_enable(0)
                order = np.lexsort((hits["slot"], hits["target"], hits["tile"]))
                hits = hits[order]
                # the bytes of the wells that figure in a duplicate come back, for the stderr log (:260-262):
                # a few hundred per tile, not the 200 000 some target touches
                for i in np.unique(hits["tile"]):
                    sel = hits[hits["tile"] == i]
                    ws = np.unique(np.concatenate([centre[sel["target"]], nbr[sel["slot"]]]).astype(np.int64))
                    seq_wells[int(i)] = ws
                    seq_bytes[int(i)] = sc.gather_wells_batch(tb, int(i), ws)
            release(tb, keep=bi + depth < len(batches))
            if overlap and bi + depth < len(batches):
                ahead.append(start_ahead(batches[bi + depth]))
            for i, t in enumerate(chunk):
                counts[(lane, t)] = report.TileCounts.from_block(blocks[i], levels)
                if want_log:
                    lines = ["Reading tile %s in lane %s" % (t, lane),
                             "Got %i sequences from %i contiguous cycle ranges." % (
                                 wells.size * want_log, want_log)]
                    sel = hits[hits["tile"] == i]
                    if sel.size:
                        # every well of the tile's duplicates decoded once, the three lines per duplicate
                        # (:260-262) put together from those strings
                        sb, sw = seq_bytes[i], seq_wells[i]
                        seqs = _decode_rows(sb)
                        cw, ww = centre[sel["target"]].astype(np.int64), nbr[sel["slot"]].astype(np.int64)
                        ci, wi = np.searchsorted(sw, cw), np.searchsorted(sw, ww)
                        for c, w, a, b, d in zip(cw.tolist(), ww.tolist(), ci.tolist(), wi.tolist(), sel["dist"].tolist()):
                            lines.append("center seq at {:>07}: {}".format(c, seqs[a]))
                            lines.append("well seq at   {:>07}: {}".format(w, seqs[b]))
                            lines.append("edit distance: {}".format(d))
                    logs[(lane, t)] = lines
            if lane_done is not None and last_batch_of[lane] == bi:
                lane_done(lane)
            if not overlap and bi + 1 < len(batches):
                ahead.append(start_ahead(batches[bi + 1]))
    finally:
        # queued loads are dropped, running ones finish - only then may their targets go
        pool.shutdown(wait=True, cancel_futures=True)
        for tb in list(live):
            release(tb)
        for tb in spare:
            tb.free()
    return counts, logs


def main(argv=None, exiting=False):
    """exiting=True (what `python -m well_duplicates_amd.count_well_duplicates` passes): the process ends
    when this returns, so the GPU context is closed without freeing its buffers one by one."""
    if os.environ.get("WD_CLI_TIMING"):
        import time
        _T0[0] = time.perf_counter()
    args = parse_args(argv)
    args.exiting = bool(exiting)
    log = (lambda msg: None) if args.quiet else (lambda msg: print(str(msg), file=sys.stderr))

    # The GPU context first (this thread's current device is its device from here on), so that the batch
    # loaders' pinned ring - pinning 64 MB takes 25 ms - can be set up by a thread of its own while this one
    # parses the targets file and lists the run directory (the call releases the GIL).  Whatever goes wrong
    # here is raised where the scanner is needed: under torchrun that is inside the ranks' failure protocol.
    from . import dist as wdist
    rank, world, local_rank = wdist.env_rank()
    device = args.device if args.device is not None else (local_rank if world > 1 else 0)
    early = {"sc": None, "err": None}
    import threading
    opener = None
    try:
        early["sc"] = Scanner(device)
        if os.environ.get("WD_INFLATE_CHUNK_MB"):            # (measurements: the pinned ring's chunk size)
            early["sc"].set_option("inflate_chunk_mb", int(os.environ["WD_INFLATE_CHUNK_MB"]))
        if not args.host_inflate:
            def warm():
                try:
                    early["sc"].set_option("inflate_warm", 1)
                except Exception as e:          # noqa: BLE001
                    early["err"] = e
            opener = threading.Thread(target=warm, name="wd-ingest-warm-up")
            opener.start()
    except Exception as e:                      # noqa: BLE001
        early["err"] = e
    try:
        return _main(args, log, wdist, rank, world, device, opener, early)
    finally:
        if opener is not None:
            opener.join()
        if early["sc"] is not None:
            early["sc"].close(exiting=args.exiting)     # (a second close is a no-op)


def _main(args, log, wdist, rank, world, device, opener, early):

    # Under torchrun the process group comes FIRST: whatever a rank's setup raises after this point - a
    # run directory it cannot list, a targets file it cannot parse, a GPU it cannot use - goes through the
    # ranks' one failure flag below, and the others end with it instead of waiting in the rendezvous for a
    # rank that has already left (bcl_direct_reader.py:59-70 and target.py:6-40 raise on the first bad path).
    if world > 1:
        import torch
        import torch.distributed as tdist
        if args.dist_backend == "nccl":
            torch.cuda.set_device(device)
            tdist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            tdist.init_process_group("gloo")            # gloo; "wd": the bootstrap channel of the unique id
    levels = args.level
    out_fh = None
    try:
        # a rank whose setup fails (an unreadable run directory, no memory on its GPU, a bad device, targets
        # it cannot upload) must not leave the others waiting in the first collective: setup and scan feed
        # ONE failure flag
        sc, err = None, None
        lanes, tiles, cycles, cycle_list, mode, k = [], [], [], [], 0, 0
        reader = csr = None
        wells = np.zeros(0, dtype=np.int64)
        n_targets = 0
        try:
            lanes = args.lane.split(",") if args.lane else range(1, 8 + 1)
            tiles = workload.tiles_for_stype(args.stype)
            if args.tile_id:
                tiles = workload.filter_tiles(tiles, args.tile_id, args.stype)
            cycles = workload.parse_cycles(args.start, args.end, args.cycles)
            cycle_list = [c for s, e in cycles for c in range(s, e)]
            mode, k = compare_mode(args.edit_distance, args.hamming)

            if args.all_wells:
                from . import cluster_indexes
                xy = cluster_indexes.read_slocs(args.slocs or os.path.join(args.run, "Data", "Intensities", "s.locs"))
            else:
                # a regular targets file is parsed in bulk; anything else goes through the reference's parser,
                # which raises what the reference raises
                fast = load_targets_csr(args.coord_file, args.level, args.sample_size)
                if fast is not None:
                    n_parsed, csr = fast[0], fast[1:]
                else:
                    targets = load_targets(filename=args.coord_file, levels=args.level + 1, limit=args.sample_size)
                    n_parsed, csr = len(targets), targets.to_csr(args.level)
                # every well some target touches = targets.get_all_indices(), sorted (the rings loaded are 1..-l)
                wells = np.unique(np.concatenate([csr[0], csr[2]]).astype(np.int64))
            run_path = args.run
            if os.environ.get("WD_TEST_RUN_SUFFIX"):        # (tests: "<rank>:<suffix>" breaks one rank's run path)
                r_, _, suffix = os.environ["WD_TEST_RUN_SUFFIX"].partition(":")
                if int(r_) == rank:
                    run_path = run_path + suffix
            reader = bcl_direct_reader.BCLReader(run_path)
            _lap("targets file, run directory")
            if args.output and rank == 0:
                out_fh = open(args.output, "w")
            if opener is not None:
                opener.join()
            if early["err"] is not None:
                raise early["err"]
            sc = early["sc"]
            if args.all_wells:
                n_targets, n_slots = sc.targets_from_coords(xy[0], xy[1], None, levels=levels)
                log("All %i wells are centres: %i neighbour slots in %i levels" % (n_targets, n_slots, levels))
                # the scan needs nothing of the rings on the host; the log of single duplicates is off
                csr = (np.zeros(0, np.int32), np.zeros((0, levels + 1), np.int32), np.zeros(0, np.int32))
            else:
                n_targets = n_parsed
                sc.set_targets(*csr)
        except Exception as e:              # noqa: BLE001 - re-raised below, on every rank
            err = e
        try:
            _lap("context, targets on the GPU")
            # (lane, tile) items are independent (count_well_duplicates.py:207-226): the flat list
            # is block-partitioned over the ranks, every rank scans its share lane by lane, and ONE
            # all-reduce of the [items, 1 + 5 levels] counter block (plus one gather of the log
            # lines) makes rank 0 hold what the single-process run holds
            lanes = list(lanes)
            items = [(lane, t) for lane in lanes for t in tiles]
            pos = {item: i for i, item in enumerate(items)}
            mine = wdist.shard(items, rank, world)
            ncnt = 1 + 5 * levels
            rows = np.zeros((len(mine), ncnt), dtype=np.int64)
            logs = {}

            def emit(lane, block):          # a finished lane: its log lines, then its report (:269)
                counts = {t: report.TileCounts.from_block(block[pos[(lane, t)]], levels) for t in tiles}
                for t in tiles:
                    lines = logs.get((lane, t))
                    if lines:
                        log("\n".join(lines))
                report.write_report(lane, n_targets, counts, verbose=not args.summary_only,
                                    strict=args.strict, out=out_fh)

            try:
                lane_tiles = [(lane, [t for (ln, t) in mine if ln == lane]) for lane in lanes] if err is None else []
                where = {item: i for i, item in enumerate(mine)}
                results = {"counts": {}, "logs": {}}      # scan_lanes fills these, lane_done reads them

                def lane_done(lane):
                    for t in dict(lane_tiles)[lane]:
                        c = results["counts"][(lane, t)]
                        rows[where[(lane, t)]] = [c.targets] + c.wells + c.dups + c.hit + c.first + c.last
                        if (lane, t) in results["logs"]:
                            logs[(lane, t)] = results["logs"][(lane, t)]
                    if world == 1:          # as the reference: a lane is reported when it is done
                        emit(lane, rows)

                if err is None:             # (a rank whose setup failed has nothing to scan: it goes to the flag)
                    scan_lanes(sc, reader, lane_tiles, cycle_list, mode, k, csr, wells,
                               max(0, args.tile_batch), args.threads,
                               0 if (args.quiet or args.all_wells) else len(cycles),
                               overlap=not args.serial_ingest,
                               interleave=resident_layout(args, mode, k, csr, reader, lanes, tiles, cycle_list),
                               gpu_inflate=not args.host_inflate, lane_done=lane_done, into=results)
            except Exception as e:          # noqa: BLE001 - re-raised below, on every rank
                err = e
            if world > 1:
                # a rank that failed must not leave the others waiting in the collective
                failed = wdist.any_rank_failed(err is not None, world, backend=args.dist_backend, device=device)
                if failed:
                    raise err if err is not None else RuntimeError("another rank failed; see its message")
                full = wdist.merge_blocks(rows, len(items), rank, world, backend=args.dist_backend,
                                          device=device, scanner=sc)
                logs = wdist.gather_dicts(logs, world)
                if rank == 0:
                    for lane in lanes:
                        emit(lane, full)
            elif err is not None:
                raise err
        finally:
            _lap("reports")
            if sc is not None:
                sc.close(exiting=args.exiting)
    finally:
        _lap("context closed")
        if out_fh:
            out_fh.close()
        if world > 1:
            import torch.distributed as tdist
            tdist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main(exiting=True))
