"""Workload construction shared by the CLI, bench.py, smoke() and the tests.

  * tile lists per sequencer type, as the reference infers them
    (count_well_duplicates.py:164-191),
  * targets CSR straight from the generator (no file round trip),
  * tile -> rank sharding (tiles are independent: count_well_duplicates.py:207-226).
"""
from __future__ import annotations

import re
from typing import List, Sequence, Tuple

import numpy as np

from . import cluster_indexes, synth

HISEQ_4000 = "hiseq_4000"
HISEQ_X = "hiseq_x"
HISEQ4000_ROWS, HISEQ4000_COLS = 2743, 1571      # SURVEY.md section 8d geometry
# NovaSeq: 4 091 904 wells per tile (cbcl_read.py:77-78).  The reference never states the grid's
# shape; 2664 x 1536 is the factorisation used here, on the same honeycomb pitch.  A lane has
# 2 surfaces x 6 swaths x 78 tiles (--stype 2678 -> 936 tile ids).
NOVASEQ_ROWS, NOVASEQ_COLS = 2664, 1536
NOVASEQ_STYPE = "2678"


def tiles_for_stype(stype: str) -> List[str]:
    """Expected tile ids of one lane (count_well_duplicates.py:166-182).

    hiseq_x -> 24 tiles x swaths 11,12,21,22 (96); hiseq_4000 -> 28 tiles (112); a number is
    the highest tile id: %100 tiles per swath, //100 = surfaces*10 + swaths.
    """
    max_tile, max_swath = 24, 22
    if stype == HISEQ_4000:
        max_tile = 28
    else:
        try:
            max_tile = int(stype) % 100
            max_swath = int(stype) // 100 or 22
        except ValueError:
            pass
    tiles = []
    for swath in ["{}{}".format(s, n) for s in range(1, max_swath // 10 + 1)
                  for n in range(1, max_swath % 10 + 1)]:
        for tile in range(1, max_tile + 1):
            tiles.append("%s%02d" % (swath, tile))
    return tiles


def filter_tiles(tiles: Sequence[str], tile_id: str, stype: str) -> List[str]:
    """-t: comma list of regexes, each anchored ^...$ (count_well_duplicates.py:185-191).

    A pattern matching nothing is an AssertionError (what the reference intends; its own
    assert message raises NameError instead - SURVEY.md section 0)."""
    out = []
    for tpat in tile_id.split(","):
        t_match = [t for t in tiles if re.match("^" + tpat + "$", t)]
        assert t_match, "%s matches no tile identifiers for a %s" % (tpat, stype)
        out.extend(t_match)
    return sorted(set(out))


def parse_cycles(start: int, end: int, cycles: str | None) -> List[Tuple[int, int]]:
    """-x/-y or --cycles a-b,c-d (count_well_duplicates.py:194-197): half-open, 0-based."""
    if cycles:
        return [(int(s), int(e)) for r in cycles.split(",") for s, e in (r.split("-"),)]
    return [(start, end)]


def targets_to_csr(targets):
    """[(centre, [ring arrays])] (cluster_indexes.generate) -> (centre, lvl_off, nbr) int32."""
    T = len(targets)
    levels = len(targets[0][1]) if T else 0
    centre = np.array([c for c, _ in targets], dtype=np.int32)
    lens = np.array([[len(r) for r in rings] for _, rings in targets], dtype=np.int64).reshape(T, levels)
    lvl_off = np.zeros((T, levels + 1), dtype=np.int64)
    lvl_off[:, 1:] = np.cumsum(lens, axis=1)
    starts = np.concatenate([[0], np.cumsum(lens.sum(axis=1))[:-1]]) if T else np.zeros(0, np.int64)
    lvl_off += starts[:, None]
    nbr = (np.concatenate([np.concatenate(rings) for _, rings in targets]).astype(np.int32)
           if T and levels else np.zeros(0, np.int32))
    return centre, lvl_off.astype(np.int32), nbr


def honeycomb_targets(rows: int, cols: int, n_targets: int, levels: int, seed=13):
    """Targets as `prepare_cluster_indexes.py -n n_targets -s seed` would emit them for a
    rows x cols honeycomb s.locs (levels > 5: rings continue at the reference's pitch)."""
    x, y = synth.honeycomb_pixels(rows, cols)
    centres = cluster_indexes.sample_centres(rows * cols, n_targets, seed)
    return targets_to_csr(cluster_indexes.generate(x, y, centres, levels))


def shard(items: Sequence, rank: int, world: int) -> List:
    """Contiguous block partition of the (lane, tile) list over ranks."""
    n = len(items)
    lo = (n * rank) // world
    hi = (n * (rank + 1)) // world
    return list(items[lo:hi])
