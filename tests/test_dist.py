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
