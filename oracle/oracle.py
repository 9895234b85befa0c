"""Python face of the CPU oracle.

TEST INFRASTRUCTURE ONLY - imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by well_duplicates_amd/.  Two restatements of the reference path:

  * `count_tile` / `tally_tile`: ctypes calls into oracle/welldup_oracle.c (plain C).
  * `py_*`: a pure-Python restatement with the reference's own structure (dict of
    sequence strings, triple loop, per-target (tally, length) tuples, text report), used
    for small cases, for the report text and as the "faithful Python" timing baseline.

Each function cites the reference lines it follows (/root/reference).
"""
from __future__ import annotations

import ctypes
import io
import os
import subprocess
from typing import Dict, List, Sequence, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libwelldup_oracle.so")

MODE_EQ, MODE_HAMMING, MODE_LEVENSHTEIN = 0, 1, 2
ERR_INDEX, ERR_EMPTY = -2, -3

_lib = None


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (make -C oracle)."""
    src = os.path.join(HERE, "welldup_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-s", "-B"])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(LIB_PATH)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        i32p = ctypes.POINTER(ctypes.c_int32)
        L.wdo_count_tile.restype = ctypes.c_int
        L.wdo_count_tile.argtypes = [
            ctypes.POINTER(u8p), ctypes.c_int, u8p, ctypes.c_int64,
            ctypes.c_int, ctypes.c_int, i32p, i32p, i32p, ctypes.c_int, ctypes.c_int,
            i32p, i32p, i32p, u8p]
        L.wdo_tally_tile.restype = None
        L.wdo_tally_tile.argtypes = [ctypes.c_int, ctypes.c_int, u8p, i32p, i32p,
                                     ctypes.POINTER(ctypes.c_int64)]
        L.wdo_hamming.restype = ctypes.c_int
        L.wdo_hamming.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
        L.wdo_levenshtein.restype = ctypes.c_int
        L.wdo_levenshtein.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
        L.wdo_count_tiles_mt.restype = ctypes.c_int
        L.wdo_count_tiles_mt.argtypes = [
            ctypes.c_int, ctypes.POINTER(u8p), ctypes.c_int, ctypes.POINTER(u8p), ctypes.c_int64,
            ctypes.c_int, ctypes.c_int, i32p, i32p, i32p, ctypes.c_int, ctypes.c_int,
            ctypes.POINTER(ctypes.c_int64), ctypes.c_int]
        _lib = L
    return _lib


def _ptr(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def hamming(a: str, b: str) -> int:
    """Levenshtein.hamming (count_well_duplicates.py:200): equal-length strings only."""
    if len(a) != len(b):
        raise ValueError("hamming: strings of unequal length")
    return lib().wdo_hamming(a.encode(), b.encode(), len(a))


def levenshtein(a: str, b: str) -> int:
    """Levenshtein.distance (count_well_duplicates.py:200)."""
    return lib().wdo_levenshtein(a.encode(), len(a), b.encode(), len(b))


def count_tile(planes: Sequence[np.ndarray], filt: np.ndarray, centre, lvl_off, nbr,
               mode: int, k: int, want_dist: bool = False):
    """count_well_duplicates.py:228-265 for one tile, through the C oracle.

    planes: L uint8 arrays of N bytes; filt: N uint8; CSR as AllTargets.to_csr().
    Returns (valid[T] uint8, dups[T, levels] int32 (-1 = invalid centre),
             lens[T, levels] int32, dist[P] int32 or None).
    Raises IndexError / AssertionError as the reference would.
    """
    planes = [np.ascontiguousarray(p, dtype=np.uint8) for p in planes]
    filt = np.ascontiguousarray(filt, dtype=np.uint8)
    centre = np.ascontiguousarray(centre, dtype=np.int32)
    lvl_off = np.ascontiguousarray(lvl_off, dtype=np.int32)
    nbr = np.ascontiguousarray(nbr, dtype=np.int32)
    T = centre.shape[0]
    levels = lvl_off.shape[1] - 1
    N = filt.shape[0]
    for p in planes:
        assert p.shape[0] == N      # bcl_direct_reader.py:338: plane length == filter length
    L = len(planes)
    pp = (ctypes.POINTER(ctypes.c_uint8) * max(L, 1))(*[_ptr(p, ctypes.c_uint8) for p in planes])
    dups = np.zeros((T, levels), dtype=np.int32)
    lens = np.zeros((T, levels), dtype=np.int32)
    valid = np.zeros(T, dtype=np.uint8)
    dist = np.zeros(max(nbr.shape[0], 1), dtype=np.int32) if want_dist else None
    rc = lib().wdo_count_tile(
        pp, L, _ptr(filt, ctypes.c_uint8), N, T, levels,
        _ptr(centre, ctypes.c_int32), _ptr(lvl_off, ctypes.c_int32), _ptr(nbr, ctypes.c_int32),
        mode, k, _ptr(dups, ctypes.c_int32), _ptr(lens, ctypes.c_int32),
        _ptr(dist, ctypes.c_int32) if want_dist else None, _ptr(valid, ctypes.c_uint8))
    if rc == ERR_INDEX:
        raise IndexError("Requested cluster is out of range")       # bcl_direct_reader.py:186-192
    if rc == ERR_EMPTY:
        raise AssertionError("empty level")                          # count_well_duplicates.py:249
    if rc != 0:
        raise RuntimeError("oracle error %d" % rc)
    return valid, dups, lens, (dist[:nbr.shape[0]] if want_dist else None)


def tally_tile(valid, dups, lens) -> np.ndarray:
    """count_well_duplicates.py:63-106 -> [targets, wells[], dups[], hits[], acco[], acci[]]."""
    T, levels = dups.shape
    block = np.zeros(1 + 5 * levels, dtype=np.int64)
    lib().wdo_tally_tile(T, levels, _ptr(np.ascontiguousarray(valid, dtype=np.uint8), ctypes.c_uint8),
                         _ptr(np.ascontiguousarray(dups, dtype=np.int32), ctypes.c_int32),
                         _ptr(np.ascontiguousarray(lens, dtype=np.int32), ctypes.c_int32),
                         _ptr(block, ctypes.c_int64))
    return block


def count_tiles_mt(planes_per_tile: Sequence[Sequence[np.ndarray]], filters: Sequence[np.ndarray],
                   centre, lvl_off, nbr, mode: int, k: int, threads: int) -> np.ndarray:
    """Many tiles, OpenMP over tiles (one thread per tile at a time): the CPU baseline leg.

    Returns the [n_tiles, 1 + 5*levels] block of `tally_tile` rows.
    """
    n_tiles = len(filters)
    L = len(planes_per_tile[0]) if n_tiles else 0
    N = filters[0].shape[0] if n_tiles else 0
    centre = np.ascontiguousarray(centre, dtype=np.int32)
    lvl_off = np.ascontiguousarray(lvl_off, dtype=np.int32)
    nbr = np.ascontiguousarray(nbr, dtype=np.int32)
    T = centre.shape[0]
    levels = lvl_off.shape[1] - 1
    keep = []
    flat = []
    for tp in planes_per_tile:
        assert len(tp) == L
        for p in tp:
            p = np.ascontiguousarray(p, dtype=np.uint8)
            assert p.shape[0] == N
            keep.append(p)
            flat.append(_ptr(p, ctypes.c_uint8))
    fl = []
    for f in filters:
        f = np.ascontiguousarray(f, dtype=np.uint8)
        keep.append(f)
        fl.append(_ptr(f, ctypes.c_uint8))
    pp = (ctypes.POINTER(ctypes.c_uint8) * max(len(flat), 1))(*flat)
    fp = (ctypes.POINTER(ctypes.c_uint8) * max(len(fl), 1))(*fl)
    out = np.zeros((n_tiles, 1 + 5 * levels), dtype=np.int64)
    rc = lib().wdo_count_tiles_mt(n_tiles, pp, L, fp, N, T, levels,
                                  _ptr(centre, ctypes.c_int32), _ptr(lvl_off, ctypes.c_int32),
                                  _ptr(nbr, ctypes.c_int32), mode, k,
                                  _ptr(out, ctypes.c_int64), threads)
    if rc != 0:
        raise RuntimeError("oracle error %d" % rc)
    return out


# ---------------------------------------------------------------------------------------
# Pure-Python restatement (reference structure)
# ---------------------------------------------------------------------------------------

def py_hamming(a: str, b: str) -> int:
    if len(a) != len(b):
        raise ValueError("hamming: strings of unequal length")
    return sum(x != y for x, y in zip(a, b))


def py_levenshtein(a: str, b: str) -> int:
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j - 1] + (ca != cb), prev[j] + 1, cur[j - 1] + 1))
        prev = cur
    return prev[-1]


def py_get_seqs(planes: Sequence[bytes], filt: bytes, cluster_indices) -> Dict[int, Tuple[str, bool]]:
    """Tile.get_seqs (bcl_direct_reader.py:158-220) on already-decompressed planes.

    planes: one bytes object per cycle of the wanted range; filt: the .filter payload.
    """
    n = len(filt)
    seq_collector = {int(idx): ["N"] * len(planes) for idx in cluster_indices}   # :181
    sorted_keys = sorted(seq_collector.keys())
    if sorted_keys and sorted_keys[-1] >= n:                                       # :186
        raise IndexError("Requested cluster %i is out of range.  Highest on this tile is %i."
                         % (sorted_keys[-1], n - 1))
    if sorted_keys and sorted_keys[0] < 0:                                         # :191
        raise IndexError("Requested cluster %i is a negative number." % sorted_keys[0])
    flags = {idx: bool(filt[idx] & 1) for idx in sorted_keys}                      # :195-197,:246
    for cyc, plane in enumerate(planes):                                           # :200
        assert len(plane) == n                                                     # :338
        for idx in sorted_keys:                                                    # :347
            b = plane[idx]
            if b:                                                                  # :352
                seq_collector[idx][cyc] = "ACGT"[b & 3]                            # :354
    return {idx: ("".join(s), flags[idx]) for idx, s in seq_collector.items()}     # :220


def py_count_tile(targets: List[List[List[int]]], seqs: Dict[int, Tuple[str, bool]],
                  levels: int, metric, edit_distance: int, log=None):
    """count_well_duplicates.py:228-265: returns the tile's list of per-target stats.

    targets: [[centre], [ring 1 ...], ...] per target in file order.
    metric: callable(a, b) -> int (py_hamming / py_levenshtein / the C versions).
    """
    tile_stats = []
    for coords in targets:                                                         # :228
        center = coords[0][0]
        if not seqs[center][1]:                                                    # :236
            continue
        center_seq = seqs[center][0]
        target_stats = [None] * levels
        tile_stats.append(target_stats)
        for level in range(levels):                                                # :244
            dups = 0
            well_indices = list(coords[level + 1])
            assert len(well_indices) > 0                                           # :249
            for well_index in well_indices:
                well_seq = seqs[well_index][0]
                dist = metric(center_seq, well_seq)                                # :252
                if dist <= edit_distance:                                          # :258
                    dups += 1
                    if log is not None:                                            # :260-262
                        log("center seq at {:>07}: {}".format(center, center_seq))
                        log("well seq at   {:>07}: {}".format(well_index, well_seq))
                        log("edit distance: {}".format(dist))
            target_stats[level] = (dups, len(well_indices))                        # :265
    return tile_stats


def py_output_writer(lane, sample_size, lane_dupl, levels=0, verbose=False) -> str:
    """output_writer (count_well_duplicates.py:27-153) returning the text it would print."""
    out = io.StringIO()
    p = lambda *a: print(*a, file=out)
    if not levels:                                                                 # :41-47
        for atile in lane_dupl.values():
            if len(atile) > 0:
                levels = len(atile[0])
                break
    tot_targets = 0
    tot_wells = [0] * levels
    tot_dups = [0] * levels
    tot_hits = [0] * levels
    tot_acco = [0] * levels
    tot_acci = [0] * levels
    for tile in sorted(lane_dupl.keys()):                                          # :63
        tile_counts = lane_dupl[tile]
        targets = len(tile_counts)
        tot_targets += targets
        if verbose:
            p("Lane: %s\tTile: %s\tTargets: %i/%i" % (lane, tile, targets, sample_size))
        acco = [0] * levels
        acci = [0] * levels
        for targ in tile_counts:                                                   # :79-89
            seen_hit = 0
            for lev in range(levels):
                if targ[lev][0]:
                    seen_hit = 1
                acco[lev] += seen_hit
            seen_hit = 0
            for lev in reversed(range(levels)):
                if targ[lev][0]:
                    seen_hit = 1
                acci[lev] += seen_hit
        for lev in range(levels):                                                  # :91-106
            wells = sum(targ[lev][1] for targ in tile_counts)
            dups = sum(targ[lev][0] for targ in tile_counts)
            hits = sum(bool(targ[lev][0]) for targ in tile_counts)
            if verbose:
                p("Level: %i\tWells: %i\tDups: %i\tHit: %i\tAccO: %i\tAccI: %i" % (
                    lev + 1, wells, dups, hits, acco[lev], acci[lev]))
            tot_wells[lev] += wells
            tot_dups[lev] += dups
            tot_hits[lev] += hits
            tot_acco[lev] += acco[lev]
            tot_acci[lev] += acci[lev]
    if tot_acci:                                                                   # :111-125
        grand_tot_hits = tot_acci[0]
        grand_tot_dups = sum(tot_dups)
        peds = (grand_tot_hits * (1 - grand_tot_hits / (grand_tot_dups + grand_tot_hits)) /
                tot_targets)
        peds2 = (grand_tot_hits * (1 - grand_tot_hits / (2 * grand_tot_dups)) / tot_targets)
    else:
        grand_tot_hits = peds = peds2 = 0
    p("LaneSummary: %s\tTiles: %i\tTargets: %i/%i" % (
        lane, len(lane_dupl), tot_targets, sample_size * len(lane_dupl)))
    for lev in range(levels):                                                      # :134-146
        p("Level: %i\tWells: %i\tDups: %i (%.5f)\t" % (
            lev + 1, tot_wells[lev], tot_dups[lev], tot_dups[lev] / tot_wells[lev]) +
          "Hit: %i (%.5f)\tAccO: %i (%.5f)\tAccI: %i (%.5f)" % (
              tot_hits[lev], tot_hits[lev] / tot_targets,
              tot_acco[lev], tot_acco[lev] / tot_targets,
              tot_acci[lev], tot_acci[lev] / tot_targets))
    raw_dup_rate = grand_tot_hits / tot_targets if grand_tot_hits else 0.0         # :148
    p()
    p("Overall duplication (Acc/Targets): {:.2%}".format(raw_dup_rate))
    p("Picard-equivalent duplication v1:  {:.2%}".format(peds))
    p("Picard-equivalent duplication v2:  {:.2%}".format(peds2))
    return out.getvalue()


def py_load_targets(filename, levels=None, limit=None) -> List[List[List[int]]]:
    """target.py:6-40 restated: returns [[centre], ring1, ...] per target, file order.

    Raises ValueError / AssertionError exactly where the reference does.
    """
    out: List[List[List[int]]] = []
    seen = set()
    nlev = None
    targ_lines = None
    with open(filename) as fh:
        lines = [x.rstrip() for x in fh] + [""]                                   # :25
    for aline in lines:
        if "," not in aline:                                                       # :27
            if targ_lines:
                coords = [[int(x) for x in l.split(",")] for l in targ_lines[:levels]]
                assert len(coords[0]) == 1                                         # :113
                assert coords[0][0] not in seen                                    # :72
                if nlev is None:
                    nlev = len(coords)
                else:
                    assert nlev == len(coords)                                     # :78
                seen.add(coords[0][0])
                out.append(coords)
                if limit and len(out) == limit:                                    # :33
                    break
            targ_lines = []
        targ_lines.append(aline)
    return out
