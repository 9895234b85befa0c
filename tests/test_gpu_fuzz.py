"""Randomised parity sweep on the GPU: small random tiles, ragged targets with ring sizes
around every pass / queue / group boundary, every mode and kernel family, random tunables -
all against the CPU oracle.  Fixed seeds, so failures reproduce."""
import numpy as np
import pytest

from helpers import blocks_to_reference, compact_tile
from oracle import oracle
from well_duplicates_amd import synth
from well_duplicates_amd.scanner import INVALID_TARGET, Scanner, TileBatch

pytestmark = pytest.mark.gpu

K_CHOICES = [1, 2, 3, 31, 32, 33, 62, 63, 64, 65, 126, 127, 128, 129, 253, 254, 255, 300, 507, 508, 509, 700]


def _targets(rng, n_clusters, T, levels, kmode):
    centres = rng.choice(n_clusters, size=T, replace=False)
    lvl_off = np.zeros((T, levels + 1), dtype=np.int32)
    nbr, pos = [], 0
    for t, c in enumerate(centres):
        if kmode == "boundary":
            K = int(rng.choice(K_CHOICES))
        elif kmode == "tiny":
            K = int(rng.integers(levels, levels + 6))
        else:
            K = int(rng.integers(levels, 140))
        K = max(K, levels)
        cuts = np.sort(rng.choice(np.arange(1, K), size=levels - 1, replace=False)) if levels > 1 else []
        sizes = np.diff(np.concatenate([[0], cuts, [K]])).astype(int)
        lvl_off[t, 0] = pos
        for l, n in enumerate(sizes):
            cand = c + rng.integers(-6, 7, size=n) + rng.integers(-3, 4, size=n) * 61
            nbr.extend(np.clip(cand, 0, n_clusters - 1).tolist())
            pos += int(n)
            lvl_off[t, l + 1] = pos
    return centres.astype(np.int32), lvl_off, np.asarray(nbr, dtype=np.int32)


# a few fixed shapes that hit the pass (127), queue (256) and group (64 / 256) boundaries, then
# random ones
FIXED = {0: (4099, 50, 5, 257, "boundary"), 1: (20011, 64, 3, 400, "boundary"), 2: (1000, 9, 7, 255, "tiny"),
         3: (20011, 17, 2, 65, "mixed"), 4: (4099, 3, 1, 256, "boundary")}


import os
N_SEEDS = int(os.environ.get("WD_FUZZ_SEEDS", "14"))      # raise for a soak run


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_fuzz(seed):
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.choice([257, 1000, 4099, 20011]))
    L = int(rng.choice([1, 2, 3, 4, 7, 8, 9, 15, 16, 17, 31, 50, 63, 64, 65, 100]))
    levels = int(rng.integers(1, 13))            # (9+: beyond the 8-bit hit masks of the line walk and the dense path)
    T = int(rng.choice([1, 2, 63, 64, 65, 100, 255, 256, 257, 400])) if n > 500 else int(rng.integers(1, 200))
    T = min(T, n)
    kmode = str(rng.choice(["boundary", "tiny", "mixed"]))
    if seed in FIXED:
        n, L, levels, T, kmode = FIXED[seed]
    print("fuzz seed %d: n=%d L=%d levels=%d T=%d %s" % (seed, n, L, levels, T, kmode))
    centre, lvl_off, nbr = _targets(rng, n, T, levels, kmode)
    spec = synth.SynthSpec(seed=100 + seed, n_clusters=n, row=61,
                           plant_per_64k=int(rng.choice([0, 3000, 30000, 65536])),
                           nocall_per_64k=int(rng.choice([0, 300, 5000, 65536])),
                           pass_per_64k=int(rng.choice([65536, 45000, 20000])), plant_far=bool(rng.integers(0, 2)))
    tiles = [(1, 1101), (2, 1203), (3, 2101)]
    cycles = list(range(2, 2 + L))
    with Scanner(0) as sc:
        sc.set_option("sort_strip", int(rng.choice([0, 8, 16, 30, 256])))
        sc.set_targets(centre, lvl_off, nbr)
        tb = TileBatch(sc, len(tiles), L, n)
        tb.fill_synthetic(spec, tiles, cycles)
        il = TileBatch(sc, len(tiles), L, n, interleave=4)          # the layout the CLI keeps resident where it is served
        il.fill_synthetic(spec, tiles, cycles)
        slots_max = int((lvl_off[:, -1] - lvl_off[:, 0]).max())
        host = [compact_tile(spec, lane, tile, cycles, centre, nbr) for lane, tile in tiles]
        modes = [(0, 0), (1, int(rng.integers(0, 5))), (1, int(rng.integers(-1, L + 2))),
                 (2, 2), (2, int(rng.integers(2, 8))), (2, int(rng.integers(8, 30)))]
        for mode, k in modes:
            want = []
            for planes, filt, c2, n2, _ in host:
                valid, dups, lens, _ = oracle.count_tile(planes, filt, c2, lvl_off, n2, mode, k)
                want.append((np.where(valid[:, None] == 1, dups, -1), oracle.tally_tile(valid, dups, lens)))
            for trial in range(3):
                opts = {"targets_per_block": int(rng.choice([1, 3, 16, 64])),
                        "queue_kernel": int(rng.integers(0, 2)) if trial else 1,
                        "dense_kernel": int(rng.choice([-1, 0, 1])) if trial else -1,
                        "dense_pack": int(rng.choice([-1, 0, 1])) if trial else -1,
                        "early_exit": int(rng.integers(0, 2)) if trial == 2 else 1,
                        "queue_first": int(rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 8])),
                        # round 3: the walk order of the queue kernels (the strips were laid out when the
                        # targets were set, with the row length 61 read off the neighbour lists), the closed
                        # form or the banded DP for Levenshtein <= 2, the dense scan in parts
                        "sort_targets": int(rng.integers(0, 2)) if trial else 1,
                        "lev2_closed": int(rng.integers(0, 2)) if trial else 1,
                        "dense_overlap": int(rng.integers(0, 2)) if trial else 0,
                        "dense_part_tiles": int(rng.choice([0, 1, 2])),
                        # the pairs walked in the order of their neighbour wells (k_scan_lines), in blocks
                        # of a few hundred to a few thousand pairs
                        "line_walk": int(rng.choice([-1, 0, 1, 1])) if trial else -1,
                        "line_pairs": int(rng.choice([0, 130, 700, 5000]))}
                for name, v in opts.items():
                    sc.set_option(name, v)
                try:
                    blocks, pt = tb.count(mode, k, per_target=True)
                except RuntimeError as e:          # the one documented limit: LDS of the generic DP
                    assert "64 KB" in str(e) and mode == 2 and k >= 18
                    continue
                for i in range(len(tiles)):
                    got = pt[i].astype(np.int64)
                    got[got == INVALID_TARGET] = -1
                    assert (got == want[i][0]).all(), (seed, mode, k, opts, i)
                    assert (blocks_to_reference(blocks[i], levels) == want[i][1]).all(), (seed, mode, k, opts)
                # the same trial on the interleaved batch, where the kernels that read it serve the mode
                # (k_scan_q<.., 4>: targets of at most 508 slots; k_scan_lines<.., 4> when the walk is on and applies)
                served = mode == 0 or (mode == 1 and k <= 254) or (mode == 2 and k <= 3)
                if served and opts["queue_kernel"] and opts["early_exit"] and opts["dense_kernel"] != 1:
                    try:
                        blocks, pt = il.count(mode, k, per_target=True)
                    except RuntimeError as e:       # the one documented refusal: nobody reads this layout for these targets
                        assert "interleaved layout" in str(e) and slots_max > 508, (seed, mode, k, opts, str(e))
                        continue
                    for i in range(len(tiles)):
                        got = pt[i].astype(np.int64)
                        got[got == INVALID_TARGET] = -1
                        assert (got == want[i][0]).all(), ("interleaved", seed, mode, k, opts, i, sc.last_kernel())
                        assert (blocks_to_reference(blocks[i], levels) == want[i][1]).all(), ("interleaved", seed, mode, k, opts)
        tb.free()
        il.free()


# ---- the dense path's window groups -------------------------------------------------------
# Consecutive centres on a small grid with a perturbed ring pattern: wells missing from some
# targets' rings (membership bits), an offset that sits in different rings for different targets
# ((ring, offset) union), tile edges (union elements that point outside the tile for some lanes),
# a ring listed out of order or a well listed twice (the group must fall back to the gathers),
# a partial last group, several tile chunks, tiny survivor queues (overflow paths).
N_WIN_SEEDS = int(os.environ.get("WD_FUZZ_WINDOW_SEEDS", "10"))


def _grid_targets(rng, rows, cols, c_first, T, levels, n_off, p_drop, p_flip, p_mess, symmetric=False):
    offs = set()
    while len(offs) < n_off:
        dr, dc = int(rng.integers(-3, 4)), int(rng.integers(-7, 8))
        if (dr, dc) != (0, 0):
            offs.add((dr, dc))
            if symmetric:                                  # b in a's ring r <=> a in b's ring r, as rings cut out of distances are
                offs.add((-dr, -dc))
    offs = sorted(offs)
    base_level = {o: int(rng.integers(0, levels)) for o in offs}
    if symmetric:
        for o in offs:
            base_level[(-o[0], -o[1])] = base_level[o]
    n = rows * cols
    centre = np.arange(c_first, c_first + T, dtype=np.int32)
    lvl_off = np.zeros((T, levels + 1), dtype=np.int32)
    nbr, pos = [], 0
    for t, c in enumerate(centre):
        r0, c0 = divmod(int(c), cols)
        rings = [[] for _ in range(levels)]
        for (dr, dc) in offs:
            r, cc = r0 + dr, c0 + dc
            if not (0 <= r < rows and 0 <= cc < cols) or rng.random() < p_drop:
                continue
            lev = base_level[(dr, dc)]
            if rng.random() < p_flip:
                lev = int(rng.integers(0, levels))
            rings[lev].append(r * cols + cc)
        lvl_off[t, 0] = pos
        for l in range(levels):
            ring = sorted(set(rings[l]))
            if not ring:                                   # the reference asserts non-empty rings (:249)
                ring = [int(c) + 1 if int(c) + 1 < n else int(c) - 1]
            if rng.random() < p_mess:
                if rng.random() < 0.5 and len(ring) > 1:
                    ring = ring[::-1]                      # not ascending
                else:
                    ring = ring + [ring[0]]                # a well twice
            nbr.extend(ring)
            pos += len(ring)
            lvl_off[t, l + 1] = pos
    return centre, lvl_off, np.asarray(nbr, dtype=np.int32)


@pytest.mark.parametrize("seed", range(N_WIN_SEEDS))
def test_fuzz_window_groups(seed):
    rng = np.random.default_rng(7000 + seed)
    rows, cols = int(rng.integers(8, 30)), int(rng.choice([64, 70, 97, 128, 200, 257]))
    n = rows * cols
    levels = int(rng.integers(1, 6))
    L = int(rng.choice([3, 9, 10, 11, 16, 17, 24, 40, 150]))
    T = int(rng.choice([n, n - 1, n // 2 + 7, 64, 65, 129]))
    T = max(1, min(T, n))
    c_first = int(rng.integers(0, n - T + 1))
    p_mess, p_flip = float(rng.choice([0.0, 0.0, 0.003])), float(rng.choice([0.0, 0.05]))
    p_drop = float(rng.choice([0.0, 0.02, 0.3]))
    # every third seed: every well a centre and a symmetric neighbour relation - the dense path then compares
    # each pair from its lower well only and records a duplicate at both ends (option dense_sym)
    symmetric = seed % 3 == 1
    if symmetric:
        T, c_first, p_mess, p_flip, p_drop = n, 0, 0.0, 0.0, 0.0
    centre, lvl_off, nbr = _grid_targets(rng, rows, cols, c_first, T, levels, int(rng.choice([4, 12, 36, 60, 90])),
                                         p_drop, p_flip, p_mess, symmetric)
    pairs = {}
    for t in range(T):
        for l in range(levels):
            for w in nbr[lvl_off[t, l]:lvl_off[t, l + 1]]:
                pairs[(int(centre[t]), int(w))] = pairs.get((int(centre[t]), int(w)), ()) + (l,)
    # what k_dense_symcheck establishes: consecutive centres, every neighbour a centre, rings in strictly
    # ascending order, b in ring r of a exactly when a in ring r of b
    ascending = all((np.diff(nbr[lvl_off[t, l]:lvl_off[t, l + 1]]) > 0).all() for t in range(T) for l in range(levels))
    is_sym = T >= 2 and ascending and all(c_first <= w < c_first + T and w != c and pairs.get((w, c)) == v
                                          for (c, w), v in pairs.items())
    spec = synth.SynthSpec(seed=300 + seed, n_clusters=n, row=cols,
                           plant_per_64k=int(rng.choice([0, 2000, 20000, 65536])),
                           nocall_per_64k=int(rng.choice([0, 300, 5000])),
                           pass_per_64k=int(rng.choice([65536, 45000, 20000])), plant_far=bool(rng.integers(0, 2)))
    n_tiles = int(rng.choice([1, 3, 5]))
    tiles = [(1 + i % 3, 1101 + i) for i in range(n_tiles)]
    cycles = list(range(1, 1 + L))
    print("window fuzz seed %d: %dx%d T=%d first=%d levels=%d L=%d tiles=%d" % (seed, rows, cols, T, c_first, levels, L, n_tiles))
    with Scanner(0) as sc:
        sc.set_targets(centre, lvl_off, nbr)
        tb = TileBatch(sc, n_tiles, L, n)
        tb.fill_synthetic(spec, tiles, cycles)
        host = [([synth.plane_bytes(spec, lane, tile, c) for c in cycles], synth.filter_bytes(spec, lane, tile))
                for lane, tile in tiles]
        try:
            for mode, k in ((0, 0), (1, 1), (1, 2), (2, 2)):
                want = []
                for planes, filt in host:
                    valid, dups, lens, _ = oracle.count_tile(planes, filt, centre, lvl_off, nbr, mode, k)
                    want.append((np.where(valid[:, None] == 1, dups, -1), oracle.tally_tile(valid, dups, lens)))
                for trial in range(3):
                    opts = {"dense_kernel": 1,
                            "dense_windows": int(rng.integers(0, 2)) if trial else 1,
                            "dense_tile_chunk": int(rng.choice([1, 2, 3, 8, 16])),
                            "dense_queue_cap": int(rng.choice([0, 0, 1, 3, 64])),
                            "dense_pack": int(rng.choice([-1, 0, 1])) if trial else -1,
                            "dense_overlap": int(rng.integers(0, 2)) if trial else 0,
                            "dense_part_tiles": int(rng.choice([0, 1, 2, 3])),
                            "dense_pack_blocks": int(rng.choice([0, 7, 1024])),
                            "dense_sym": int(rng.integers(0, 2)) if trial else 1}
                    for name, v in opts.items():
                        sc.set_option(name, v)
                    blocks, pt = tb.count(mode, k, per_target=True)
                    assert sc.get_option("dense_sym_on") == (1 if is_sym and opts["dense_sym"] else 0), (seed, opts, is_sym)
                    # clean input (with ring flips a union of (ring, offset) pairs may outgrow the 128
                    # elements a window group holds): the path under test is the one that ran
                    if p_mess == 0.0 and p_flip == 0.0 and T >= 128:
                        assert sc.get_option("dense_window_groups") >= (T // 64) // 2, sc.get_option("dense_window_groups")
                    for i in range(n_tiles):
                        got = pt[i].astype(np.int64)
                        got[got == INVALID_TARGET] = -1
                        assert (got == want[i][0]).all(), (seed, mode, k, opts, i)
                        assert (blocks_to_reference(blocks[i], levels) == want[i][1]).all(), (seed, mode, k, opts)
        finally:
            tb.free()
