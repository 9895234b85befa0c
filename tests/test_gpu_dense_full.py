"""BASELINE configs[4] at its own size: every well of a full tile a centre (4 309 253 centres,
3 levels, device-generated rings), 150 bp, 2 % planted duplicates - the dense path
(csrc/scan_dense.inc) against the CPU oracle on sampled centres and against the queue kernel on
the whole tile, in the three compare modes.  Reference loop being replaced:
count_well_duplicates.py:228-265."""
import numpy as np
import pytest

from helpers import blocks_to_reference, compact_tile
from oracle import oracle
from well_duplicates_amd import synth, workload
from well_duplicates_amd.scanner import INVALID_TARGET, Scanner, TileBatch

pytestmark = pytest.mark.gpu

ROWS, COLS, LEVELS, L = workload.HISEQ4000_ROWS, workload.HISEQ4000_COLS, 3, 150
MODES = ((0, 0, "equality"), (1, 2, "hamming<=2"), (2, 2, "levenshtein<=2"))


@pytest.fixture(scope="module")
def setup():
    sc = Scanner(0)
    n = ROWS * COLS
    x, y = synth.honeycomb_pixels(ROWS, COLS)
    T, P = sc.targets_from_coords(x, y, None, levels=LEVELS)
    assert T == n == 4309253
    spec = synth.SynthSpec(seed=5, n_clusters=n, row=COLS, plant_per_64k=1311, nocall_per_64k=328)
    tb = TileBatch(sc, 1, L, n)
    tb.fill_synthetic(spec, [(1, 1101)], list(range(L)))
    yield sc, tb, spec, n
    tb.free()
    sc.close()


def test_dense_path_vs_oracle_on_sampled_centres(setup):
    """Per-target duplicate counts of the dense path for 3 000 centres - a random sample plus every
    centre the device reports a duplicate for among the first 200 000 - against oracle.count_tile on
    the same wells' bytes."""
    sc, tb, spec, n = setup
    centre, lvl_off, nbr = sc.get_targets()
    rng = np.random.default_rng(150)
    for mode, k, name in MODES:
        sc.set_option("dense_kernel", -1)
        blocks, pt = tb.count(mode, k, per_target=True)
        assert sc.last_kernel().startswith("dense chain"), sc.last_kernel()
        assert "pairs from one end" in sc.last_kernel()        # rings cut out of distances are symmetric: each pair compared once
        assert sc.get_option("dense_window_groups") >= (n + 63) // 64 - 1      # compared from LDS windows
        got_all = pt[0].astype(np.int64)
        got_all[got_all == INVALID_TARGET] = -1
        with_dups = np.flatnonzero((got_all[:200000] > 0).any(axis=1))[:1000]
        sample = np.unique(np.concatenate([rng.choice(n, 2000, replace=False), with_dups,
                                           [0, 1, COLS - 1, COLS, n - COLS, n - 1]]))
        # the sampled targets as a small CSR of their own, wells remapped to the bytes generated for them
        c_s = centre[sample]
        off_s = np.zeros((sample.size, LEVELS + 1), dtype=np.int64)
        parts = []
        pos = 0
        for i, t in enumerate(sample):
            off_s[i] = lvl_off[t] - lvl_off[t, 0] + pos
            parts.append(nbr[lvl_off[t, 0]:lvl_off[t, LEVELS]])
            pos = off_s[i, LEVELS]
        nbr_s = np.concatenate(parts)
        planes, filt, c2, n2, _ = compact_tile(spec, 1, 1101, list(range(L)), c_s, nbr_s)
        valid, dups, lens, _ = oracle.count_tile(planes, filt, c2, off_s.astype(np.int32), n2, mode, k)
        want = np.where(valid[:, None] == 1, dups, -1)
        assert (got_all[sample] == want).all(), name
        assert (want > 0).sum() > 300, name                        # the sample holds real duplicates
        # Wells per target are the ring sizes
        assert (lens == np.diff(off_s, axis=1)).all()


def test_dense_path_vs_queue_kernel_on_the_whole_tile(setup):
    """The tile's tally block and every per-target count: dense path = queue kernel (the kernel the
    golden fixtures pin), all three modes; and the tally block equals the tallies of the per-target
    counts (count_well_duplicates.py:80-95 restated by the oracle)."""
    sc, tb, spec, n = setup
    try:
        for mode, k, name in MODES:
            sc.set_option("dense_kernel", 0)
            want_b, want_pt = tb.count(mode, k, per_target=True)
            assert sc.last_kernel().startswith("k_scan"), sc.last_kernel()
            sc.set_option("dense_kernel", -1)
            got_b, got_pt = tb.count(mode, k, per_target=True)
            assert sc.last_kernel().startswith("dense chain")
            assert (got_b == want_b).all(), name
            assert (got_pt == want_pt).all(), name
            dups_found = int(got_b[0, 1 + LEVELS:1 + 2 * LEVELS].sum())
            assert dups_found > 50000, (name, dups_found)
            # tallies of the per-target counts, as output_writer makes them
            pt = got_pt[0].astype(np.int64)
            valid = (pt[:, 0] != INVALID_TARGET).astype(np.int32)
            _, lvl_off, _ = sc.get_targets()
            lens = np.diff(lvl_off, axis=1).astype(np.int64)
            ref = oracle.tally_tile(valid, np.where(valid[:, None] == 1, pt, 0), lens)
            assert (blocks_to_reference(got_b[0], LEVELS) == ref).all(), name
    finally:
        sc.set_option("dense_kernel", -1)


def test_part_pipeline_gives_the_same_answers(setup):
    """Option dense_overlap: the scan as a pipeline of parts over two streams and two scratch sets
    (launch_dense) - any part size, bounded or unbounded pack kernel - must give the one-chain scan's
    tally blocks, per-target counts and hit records (with scan-wide tile numbers)."""
    sc, tb1, spec, n = setup
    tiles = 5
    tb = TileBatch(sc, tiles, 40, n)
    tb.fill_synthetic(spec, [(2, 1101 + i) for i in range(tiles)], list(range(40)))
    try:
        for mode, k, name in MODES:
            sc.set_option("dense_overlap", 0)
            sc.hitlog_enable(4_000_000)
            want_b, want_pt = tb.count(mode, k, per_target=True)
            want_hits, total = sc.hitlog_fetch(4_000_000)
            assert 0 < total <= 4_000_000
            want_hits = np.sort(want_hits, order=["tile", "target", "slot"])
            assert set(np.unique(want_hits["tile"]).tolist()) == set(range(tiles))
            for part, pack_blocks in ((0, 1024), (1, 0), (2, 64), (4, 1024)):
                sc.set_option("dense_overlap", 1)
                sc.set_option("dense_part_tiles", part)
                sc.set_option("dense_pack_blocks", pack_blocks)
                got_b, got_pt = tb.count(mode, k, per_target=True)
                hits, total2 = sc.hitlog_fetch(4_000_000)
                assert (got_b == want_b).all() and (got_pt == want_pt).all(), (name, part, pack_blocks)
                assert total2 == total
                assert (np.sort(hits, order=["tile", "target", "slot"]) == want_hits).all(), (name, part)
    finally:
        sc.hitlog_enable(0)
        sc.set_option("dense_overlap", 0)
        sc.set_option("dense_part_tiles", 0)
        sc.set_option("dense_pack_blocks", 1024)
        tb.free()
