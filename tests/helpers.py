"""Shared test plumbing: golden fixtures -> inputs for the oracle and for the device."""
from __future__ import annotations

import json
import os
from functools import lru_cache

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")

from well_duplicates_amd import synth                      # noqa: E402
from well_duplicates_amd.targets import load_targets      # noqa: E402

MODE_ID = {"eq": 0, "hamming": 1, "levenshtein": 2}
FIXTURES = ["small_list", "mid", "mid_subset", "dead_tile", "seven_levels", "far", "novaseq",
            "novaseq_all_wells"]


@lru_cache(maxsize=None)
def load_fixture(name):
    with open(os.path.join(GOLD, name + ".json")) as fh:
        return json.load(fh)


@lru_cache(maxsize=None)
def fixture_targets(name):
    fx = load_fixture(name)
    t = load_targets(os.path.join(GOLD, fx["targets_file"]), levels=fx["levels"] + 1,
                     limit=fx["n_targets"])
    return t, t.to_csr(fx["levels"])


def run_cycles(run):
    return [c for a, b in run["cycles"] for c in range(a, b)]


def compact_tile(spec, lane, tile, cycles, centre, nbr, excluded_cbcl=False):
    """Planes/filter restricted to the wells the targets touch, plus remapped indices.

    The oracle only ever reads those wells, so this is equivalent to full planes and keeps
    the CPU suite fast (the synthetic bytes are a pure function of (cycle, cluster)).
    excluded_cbcl: the run stores only passing wells (NovaSeq .cbcl with the excluded flag),
    so wells that failed the filter read as no-calls (bcl_direct_reader.py:303-314)."""
    wells = np.unique(np.concatenate([centre, nbr]).astype(np.int64))
    planes = [synth.plane_bytes(spec, lane, int(tile), c, wells) for c in cycles]
    filt = synth.filter_bytes(spec, lane, int(tile), wells)
    if excluded_cbcl:
        planes = [np.where(filt & 1, p, 0).astype(np.uint8) for p in planes]
    c2 = np.searchsorted(wells, centre).astype(np.int32)
    n2 = np.searchsorted(wells, nbr).astype(np.int32)
    return planes, filt, c2, n2, wells


def lane_dupl_from(valid, dups, lens):
    """Per-target arrays -> the reference's list of [[tally, length] * levels] per valid target."""
    out = []
    for t in range(dups.shape[0]):
        if valid[t]:
            out.append([[int(dups[t, l]), int(lens[t, l])] for l in range(dups.shape[1])])
    return out


def decode_seq(planes, idx):
    return "".join("ACGT"[int(p[idx]) & 3] if p[idx] else "N" for p in planes)


def expected_dup_log(planes, centre, lvl_off, nbr, orig_centre, orig_nbr, valid, dist, k):
    """The three stderr lines per duplicate (count_well_duplicates.py:260-262)."""
    lines = []
    levels = lvl_off.shape[1] - 1
    for t in range(centre.shape[0]):
        if not valid[t]:
            continue
        cseq = decode_seq(planes, centre[t])
        for p in range(lvl_off[t, 0], lvl_off[t, levels]):
            if dist[p] <= k:
                lines.append("center seq at {:>07}: {}".format(int(orig_centre[t]), cseq))
                lines.append("well seq at   {:>07}: {}".format(int(orig_nbr[p]), decode_seq(planes, nbr[p])))
                lines.append("edit distance: {}".format(int(dist[p])))
    return lines


def blocks_to_reference(block, levels):
    """Device counter row (first/last histograms) -> oracle row (AccO/AccI columns)."""
    b = np.asarray(block, dtype=np.int64).copy()
    first = b[1 + 3 * levels: 1 + 4 * levels]
    last = b[1 + 4 * levels: 1 + 5 * levels]
    b[1 + 3 * levels: 1 + 4 * levels] = np.cumsum(first)
    b[1 + 4 * levels: 1 + 5 * levels] = np.cumsum(last[::-1])[::-1]
    return b
