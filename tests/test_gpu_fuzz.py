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
    levels = int(rng.integers(1, 8))
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
        sc.set_targets(centre, lvl_off, nbr)
        tb = TileBatch(sc, len(tiles), L, n)
        tb.fill_synthetic(spec, tiles, cycles)
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
                        "queue_first": int(rng.choice([0, 1, 2, 3, 4, 6, 8]))}
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
        tb.free()
