"""GPU parity at the limits the C ABI advertises (include/welldup.h): WD_MAX_LEVELS = 32 rings, reads as long
as a HiSeq 4000 run has cycles (308: bcl_direct_reader.py:139-142; -x/-y are unbounded), Hamming thresholds
around the queue kernels' 8-bit mismatch field (k <= 254), and the line walk fed through a pointer table.

The reference's compare loop and targets model are level-count agnostic (target.py:6-40,
count_well_duplicates.py:244-265), so every case is checked against the oracle, per target and per tile;
where a path has a dispatch edge (8-bit hit masks of the dense path and the line walk: levels <= 8; packed rows
of the dense Levenshtein: L <= 160) the test also asserts WHICH kernel ran on either side of the edge.
"""
import numpy as np
import pytest

from helpers import blocks_to_reference, compact_tile
from oracle import oracle
from well_duplicates_amd import synth
from well_duplicates_amd.scanner import INVALID_TARGET, Scanner, TileBatch

pytestmark = pytest.mark.gpu

PATHS = {"queue": {"line_walk": 0, "dense_kernel": 0},
         "lines": {"line_walk": 1, "dense_kernel": 0, "line_pairs": 900},
         "dense": {"line_walk": 0, "dense_kernel": 1}}
RESET = {"line_walk": -1, "dense_kernel": -1, "line_pairs": 0}


def _targets(rng, n_clusters, T, levels, ring, row):
    """Ragged rings close to the centre (so that planted copies are hit), repeated wells allowed."""
    centres = rng.choice(n_clusters, size=T, replace=False)
    lvl_off = np.zeros((T, levels + 1), dtype=np.int32)
    nbr, pos = [], 0
    for t, c in enumerate(centres):
        lvl_off[t, 0] = pos
        for l in range(levels):
            n = int(rng.integers(1, ring + 1))
            cand = c + rng.integers(-3, 4, size=n) + rng.integers(-2, 3, size=n) * row
            nbr.extend(np.clip(cand, 0, n_clusters - 1).tolist())
            pos += n
            lvl_off[t, l + 1] = pos
    return centres.astype(np.int32), lvl_off, np.asarray(nbr, dtype=np.int32)


def _check(sc, tb, host, lvl_off, levels, mode, k, tag):
    blocks, pt = tb.count(mode, k, per_target=True)
    for i, (planes, filt, c2, n2, _) in enumerate(host):
        valid, dups, lens, _ = oracle.count_tile(planes, filt, c2, lvl_off, n2, mode, k)
        want = np.where(valid[:, None] == 1, dups, -1)
        got = pt[i].astype(np.int64)
        got[got == INVALID_TARGET] = -1
        assert (got == want).all(), (tag, mode, k, i)
        assert (blocks_to_reference(blocks[i], levels) == oracle.tally_tile(valid, dups, lens)).all(), (tag, mode, k, i)
    return blocks


@pytest.mark.parametrize("levels", [8, 9, 16, 32])
def test_level_counts_up_to_the_abi_limit(levels):
    """8 is the last level count inside the 8-bit hit masks of the line walk and the dense path, 9 the first
    that must fall back (to k_scan_q); 16 and 32 = WD_MAX_LEVELS.  Sampled walk, forced line walk, forced
    dense path: all three against the oracle, in the three modes."""
    rng = np.random.default_rng(40 + levels)
    spec = synth.SynthSpec(seed=60 + levels, n_clusters=30011, row=97, plant_per_64k=25000,
                           nocall_per_64k=2500, pass_per_64k=50000)
    T, L = 700, 50
    centre, lvl_off, nbr = _targets(rng, spec.n_clusters, T, levels, ring=5, row=97)
    tiles = [(1, 1101), (2, 2210)]
    cycles = list(range(4, 4 + L))
    with Scanner(0) as sc:
        sc.set_targets(centre, lvl_off, nbr)
        tb = TileBatch(sc, len(tiles), L, spec.n_clusters)
        tb.fill_synthetic(spec, tiles, cycles)
        host = [compact_tile(spec, lane, tile, cycles, centre, nbr) for lane, tile in tiles]
        try:
            for path, opts in PATHS.items():
                for name, v in opts.items():
                    sc.set_option(name, v)
                for mode, k in ((0, 0), (1, 2), (2, 2), (2, 3)):
                    blocks = _check(sc, tb, host, lvl_off, levels, mode, k, (levels, path))
                    assert blocks[:, 1 + levels:1 + 2 * levels].sum() > 0            # duplicates were found
                    ran = sc.last_kernel()
                    if path == "lines" and not (mode == 2 and k == 3):
                        assert ran.startswith("k_scan_lines" if levels <= 8 else "k_scan_q"), (levels, mode, k, ran)
                    if path == "dense" and k <= 2:
                        assert ran.startswith("dense chain" if levels <= 8 else "k_scan_q"), (levels, mode, k, ran)
                    if path == "queue":
                        assert ran.startswith("k_scan_q"), ran
        finally:
            tb.free()


@pytest.mark.parametrize("L", [151, 160, 161, 200, 308])
def test_read_lengths_up_to_a_full_run(L):
    """Reads longer than any other test uses: 160 is the last length the dense path's packed rows hold
    (Levenshtein <= 2 there needs them: at 161 it hands over to the queue kernel), 308 every cycle of a
    HiSeq 4000 run.  All three modes on the three paths; at L = 308 also Hamming thresholds on both sides
    of 254 (above it the queue kernels' 8-bit mismatch count does not reach: k_scan takes over) and
    Levenshtein thresholds that use the generic LDS-row kernel."""
    rng = np.random.default_rng(500 + L)
    # heavy planting, and a low-complexity alphabet on a third of the wells would need uploaded planes:
    # the synthetic spec's no-calls and planted copies give near-duplicates at every distance instead
    spec = synth.SynthSpec(seed=80 + L, n_clusters=20011, row=97, plant_per_64k=30000,
                           nocall_per_64k=3000, pass_per_64k=52000)
    T, levels = 260, 4
    centre, lvl_off, nbr = _targets(rng, spec.n_clusters, T, levels, ring=8, row=97)
    tiles = [(1, 1101), (3, 2210)]
    cycles = list(range(0, L))
    with Scanner(0) as sc:
        sc.set_targets(centre, lvl_off, nbr)
        tb = TileBatch(sc, len(tiles), L, spec.n_clusters)
        tb.fill_synthetic(spec, tiles, cycles)
        host = [compact_tile(spec, lane, tile, cycles, centre, nbr) for lane, tile in tiles]
        try:
            for path, opts in PATHS.items():
                for name, v in opts.items():
                    sc.set_option(name, v)
                for mode, k in ((0, 0), (1, 2), (1, 40), (2, 2), (2, 3), (2, 5)):
                    _check(sc, tb, host, lvl_off, levels, mode, k, (L, path))
                    ran = sc.last_kernel()
                    if path == "dense" and mode == 2 and k == 2:
                        assert ran.startswith("dense chain" if L <= 160 else "k_scan_q"), (L, ran)
                    if path == "dense" and mode != 2 and k <= 2:
                        assert ran.startswith("dense chain"), (L, mode, k, ran)
            for name, v in RESET.items():
                sc.set_option(name, v)
            if L == 308:
                for k in (253, 254, 255, 300, 307, 308, 400):
                    for path in ("queue", "lines"):
                        for name, v in PATHS[path].items():
                            sc.set_option(name, v)
                        _check(sc, tb, host, lvl_off, levels, 1, k, (L, path, "hamming"))
                        ran = sc.last_kernel()
                        if k <= 254:
                            assert ran.startswith("k_scan_lines" if path == "lines" else "k_scan_q"), (k, ran)
                        else:                                   # the guard: beyond the 8-bit count
                            assert ran.startswith("k_scan<HamState"), (k, ran)
                for name, v in RESET.items():
                    sc.set_option(name, v)
                for k in (17, 18, 40, 253, 307, 308):           # banded DP in registers, then rows in LDS
                    _check(sc, tb, host, lvl_off, levels, 2, k, (L, "levenshtein"))
        finally:
            tb.free()


def test_line_walk_through_a_pointer_table():
    """k_scan_lines<false, ...>: planes at scrambled, odd-aligned addresses (no common stride), so the walk
    reads every plane through the pointer table - against the strided walk and the oracle."""
    rng = np.random.default_rng(77)
    spec = synth.SynthSpec(seed=78, n_clusters=30011, row=97, plant_per_64k=20000, nocall_per_64k=2000)
    T, levels, L = 600, 5, 50
    centre, lvl_off, nbr = _targets(rng, spec.n_clusters, T, levels, ring=30, row=97)
    tiles = [(1, 1101), (1, 1102), (2, 1101)]
    cycles = list(range(L))
    with Scanner(0) as sc:
        sc.set_targets(centre, lvl_off, nbr)
        tb = TileBatch(sc, len(tiles), L, spec.n_clusters)
        tb.fill_synthetic(spec, tiles, cycles)
        host = [compact_tile(spec, lane, tile, cycles, centre, nbr) for lane, tile in tiles]
        n = spec.n_clusters
        slab = sc.malloc(len(tiles) * L * (n + 13) + 64)
        try:
            ptrs = [[0] * L for _ in tiles]
            for j, o in enumerate(rng.permutation(len(tiles) * L)):
                i, c = divmod(int(o), L)
                ptrs[i][c] = slab + 1 + j * (n + 13)
                sc.h2d(ptrs[i][c], tb.download_plane(i, c))
            sc.set_option("line_walk", 1)
            sc.set_option("line_pairs", 1500)
            for mode, k in ((0, 0), (1, 1), (1, 2), (1, 3), (1, 9), (2, 2)):
                strided = _check(sc, tb, host, lvl_off, levels, mode, k, "strided walk")
                assert sc.last_kernel().startswith("k_scan_lines<true"), sc.last_kernel()
                blocks, pt = sc.count_tiles(ptrs, tb.filter_ptrs(), n, mode, k, per_target=True)
                assert sc.last_kernel().startswith("k_scan_lines<false"), sc.last_kernel()
                assert (blocks == strided).all(), (mode, k)
                for i, (planes, filt, c2, n2, _) in enumerate(host):
                    valid, dups, lens, _ = oracle.count_tile(planes, filt, c2, lvl_off, n2, mode, k)
                    got = pt[i].astype(np.int64)
                    got[got == INVALID_TARGET] = -1
                    assert (got == np.where(valid[:, None] == 1, dups, -1)).all(), (mode, k, i)
        finally:
            sc.free(slab)
            tb.free()


def test_hitlog_fetch_is_bounded_by_the_capacity():
    """A caller that asks for more records than the device log holds gets the records there are - not a
    zero-filled tail that reads as (tile 0, target 0, slot 0) hits - and sees the overflow in the total."""
    rng = np.random.default_rng(5)
    spec = synth.SynthSpec(seed=11, n_clusters=9001, row=97, nocall_per_64k=65536)        # every pair a duplicate
    centre, lvl_off, nbr = _targets(rng, spec.n_clusters, 100, 3, ring=6, row=97)
    with Scanner(0) as sc:
        sc.set_targets(centre, lvl_off, nbr)
        tb = TileBatch(sc, 1, 8, spec.n_clusters)
        tb.fill_synthetic(spec, [(1, 1101)], list(range(8)))
        try:
            sc.hitlog_enable(50)
            blocks, _ = tb.count(0, 0)
            n_dups = int(blocks[0, 1 + 3:1 + 6].sum())
            hits, total = sc.hitlog_fetch(100000)
            assert total == n_dups > 50
            assert len(hits) == 50                      # what the log held, nothing invented
            slots = set(zip(hits["target"].tolist(), hits["slot"].tolist()))
            assert len(slots) == 50                     # 50 different real records
            for t, s in slots:
                assert lvl_off[t, 0] <= s < lvl_off[t, 3]
        finally:
            sc.hitlog_enable(0)
            tb.free()
