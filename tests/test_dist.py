"""N > 1 path on CPU: world_size-2 gloo processes shard the tiles, compute their rows (the
oracle stands in for the device scan - this test is about the sharding and the collective),
all-reduce the counter block and print the report; the result must be bit-identical to the
single-process run and to the reference's golden stdout."""
import io
import os
import socket
import sys

import numpy as np
import pytest

from helpers import MODE_ID, compact_tile, fixture_targets, load_fixture, run_cycles, REPO
from oracle import oracle
from well_duplicates_amd import dist as wdist
from well_duplicates_amd import report, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _device_style_rows(name, run, items):
    """[valid, wells, dups, hit, first, last] rows as the device would emit them."""
    fx = load_fixture(name)
    spec = synth.spec_from_dict(fx["spec"])
    _, (centre, lvl_off, nbr) = fixture_targets(name)
    levels = fx["levels"]
    rows = np.zeros((len(items), 1 + 5 * levels), dtype=np.int64)
    for i, (lane, tile) in enumerate(items):
        planes, filt, c2, n2, _ = compact_tile(spec, lane, tile, run_cycles(run), centre, nbr)
        valid, dups, lens, _ = oracle.count_tile(planes, filt, c2, lvl_off, n2,
                                                 MODE_ID[run["mode"]], run["k"])
        stats = [[(int(dups[t, l]), int(lens[t, l])) for l in range(levels)]
                 for t in range(len(valid)) if valid[t]]
        tc = report.TileCounts.from_target_stats(stats, levels)
        rows[i] = [tc.targets] + tc.wells + tc.dups + tc.hit + tc.first + tc.last
    return rows


def _worker(rank, world, port, name, run_idx, out_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fx = load_fixture(name)
    run = fx["runs"][run_idx]
    items = [(lane, tile) for lane in fx["lanes"] for tile in fx["tiles"]]
    r, w, _ = wdist.env_rank()
    mine = wdist.shard(items, r, w)
    rows = _device_style_rows(name, run, mine)
    assert not wdist.any_rank_failed(False, w)
    assert wdist.any_rank_failed(rank == world - 1, w)         # one rank's failure is everybody's
    full = wdist.merge_blocks(rows, len(items), r, w, backend="gloo")
    assert isinstance(full, np.ndarray)
    logs = wdist.gather_dicts({item: ["line of rank %d" % rank] for item in mine}, w)
    assert sorted(logs) == sorted(items) and logs[items[0]] == ["line of rank 0"]
    slowest = wdist.max_over_ranks(float(rank + 1), w)
    assert slowest == float(world)
    np.save(os.path.join(out_dir, "full_%d.npy" % rank), full)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_scan_matches_single_process(tmp_path, world):
    import torch.multiprocessing as mp
    name, run_idx = "mid", 2            # 2 lanes x 2 tiles, Levenshtein <= 2
    fx = load_fixture(name)
    run = fx["runs"][run_idx]
    items = [(lane, tile) for lane in fx["lanes"] for tile in fx["tiles"]]
    port = _free_port()
    mp.spawn(_worker, args=(world, port, name, run_idx, str(tmp_path)), nprocs=world, join=True)
    single = _device_style_rows(name, run, items)
    for rank in range(world):
        full = np.load(tmp_path / ("full_%d.npy" % rank))
        assert (full == single).all()
    # the merged block prints the reference's report
    levels = fx["levels"]
    text = io.StringIO()
    targets, _ = fixture_targets(name)
    for lane in fx["lanes"]:
        tiles = {t: report.TileCounts.from_block(single[items.index((lane, t))], levels)
                 for t in fx["tiles"]}
        report.write_report(lane, len(targets), tiles, verbose=True, out=text)
    assert text.getvalue() == run["stdout"]


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 96, 896):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi = wdist.shard_bounds(n, r, world)
                seen.extend(range(lo, hi))
                assert 0 <= hi - lo <= n // world + 1
            assert seen == list(range(n))
    assert wdist.shard(list("abcdefg"), 1, 3) == ["c", "d"]


def _worker8(rank, world, port, out_dir):
    """A rank of the BASELINE configs[2] world: 8 ranks, 8 lanes x 112 tiles, 5 levels - its rows are a
    function of (lane, tile) that no other rank can produce, so the merged block shows who wrote what."""
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from well_duplicates_amd import count_well_duplicates as cwd
    from well_duplicates_amd import workload
    tiles = workload.tiles_for_stype("hiseq_4000")
    items = [(lane, t) for lane in range(1, 9) for t in tiles]
    r, w, _ = wdist.env_rank()
    mine = wdist.shard(items, r, w)
    assert len(mine) == 112 and {lane for lane, _ in mine} == {rank + 1}      # a lane per rank, as the reference's jobs
    rows = np.array([[int(lane) * 100000 + int(t) * 7 + c for c in range(26)] for lane, t in mine], dtype=np.int64)
    assert not wdist.any_rank_failed(False, w)
    assert wdist.any_rank_failed(rank == 5, w)
    full = wdist.merge_blocks(rows, len(items), r, w, backend="gloo")
    logs = wdist.gather_dicts({item: ["rank %d" % rank] for item in mine[:3]}, w)
    assert len(logs) == 24
    assert wdist.max_over_ranks(float(rank), w) == 7.0
    # eight ranks of one node share the host's cores: the reader threads are this rank's share of them
    cpus = len(os.sched_getaffinity(0))
    assert cwd.default_threads() == max(1, min(32, cpus // 8))
    np.save(os.path.join(out_dir, "full8_%d.npy" % rank), full)
    dist.barrier()
    dist.destroy_process_group()


def test_world_of_eight_at_the_configs2_shape(tmp_path):
    """BASELINE configs[2]'s process layout on CPU: 8 gloo ranks, the flat list of 8 x 112 (lane, tile) items
    block-sharded so that rank r owns lane r + 1 (Snakefile.count_dups:25, :153-160: a job per lane), ONE
    all-reduce of the [896, 26] block, the failure flag, the gather of the log lines, the reader-thread
    share at LOCAL_WORLD_SIZE = 8.  (The GPU side of this shape runs in tests/test_bench.py.)"""
    import torch.multiprocessing as mp
    from well_duplicates_amd import workload
    world = 8
    mp.spawn(_worker8, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    tiles = workload.tiles_for_stype("hiseq_4000")
    items = [(lane, t) for lane in range(1, 9) for t in tiles]
    want = np.array([[int(lane) * 100000 + int(t) * 7 + c for c in range(26)] for lane, t in items], dtype=np.int64)
    assert want.shape == (896, 26)
    for rank in range(world):
        assert (np.load(tmp_path / ("full8_%d.npy" % rank)) == want).all()
