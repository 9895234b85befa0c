"""Neighbour-index generator: the producer of the scan path's targets file.

Vectorised restatement of the reference's setup stage (prepare_cluster_indexes.py:26-78,
:99-172) with identical ring membership and identical output text, ~100x faster
(the reference costs about 70 ms per target in pure Python):

  * pixel coordinates  x = int(v * 10.0 + 1000.5)          (:110-112, truncation toward 0)
  * sample             random.seed(seed) if seed; random.sample(range(N), n)   (:26-30)
  * search window      records max(0, c - 20000) .. c + 20001 inclusive        (:52-67)
  * ring test          MAX_DISTS[lev] < dist <= MAX_DISTS[lev + 1]             (:61-63)
  * every level must be non-empty, else RuntimeError                          (:70-76)
  * output             centre line, then one comma-joined line per level      (:162-167)

Distances are compared on exact integer squares (dist <= 22  <=>  dx*dx + dy*dy <= 484:
the thresholds are integers and sqrt is monotone and exact on perfect squares), so the
membership is bit-identical to the reference's double-precision sqrt comparison.

Extensions the reference cannot do (needed for BASELINE.json configs 4 and 5): any number
of levels (rings keep the reference's 20-pixel pitch beyond level 5).
"""
from __future__ import annotations

import random
import struct
import sys
from typing import List, Sequence

import numpy as np

# prepare_cluster_indexes.py:19
MAX_DISTS = [1, 22, 42, 62, 82, 102]
MAX_SEARCH_AREA = 20000          # prepare_cluster_indexes.py:43
DEF_SAMPLE_SIZE = 2500           # prepare_cluster_indexes.py:23


def max_dists_for(levels: int) -> List[int]:
    """Ring boundaries for `levels` rings: the reference's table, continued at +20 per ring."""
    d = list(MAX_DISTS)
    while len(d) < levels + 1:
        d.append(d[-1] + 20)
    return d[:levels + 1]


def read_slocs(path: str):
    """s.locs -> (x, y) int64 pixel arrays as the reference decodes them (:135-137, :110-112)."""
    with open(path, "rb") as fh:
        head = fh.read(12)
        _, _, n = struct.unpack("=ifI", head)
        body = np.frombuffer(fh.read(8 * n), dtype="<f4")
    if body.size != 2 * n:
        raise ValueError("s.locs is truncated: header says %d records" % n)
    xy = body.reshape(n, 2).astype(np.float64)
    # int() truncates toward zero; so does the float64 -> int64 cast
    x = (xy[:, 0] * 10.0 + 1000.5).astype(np.int64)
    y = (xy[:, 1] * 10.0 + 1000.5).astype(np.int64)
    return x, y


def sample_centres(n_clusters: int, sample_size: int, seed=None) -> List[int]:
    """get_random_array (:26-30).  A falsy seed (None or 0) leaves the generator unseeded."""
    if seed:
        random.seed(seed)
    return random.sample(range(n_clusters), sample_size)


def rings_for(c: int, x: np.ndarray, y: np.ndarray, levels: int = 5,
              max_dists: Sequence[int] | None = None) -> List[np.ndarray]:
    """get_indexes (:38-78) for one centre: list of `levels` ascending index arrays."""
    md = np.asarray(max_dists if max_dists is not None else max_dists_for(levels), dtype=np.int64)
    n = x.shape[0]
    lo = max(0, c - MAX_SEARCH_AREA)
    hi = min(n, c + MAX_SEARCH_AREA + 2)       # the record at c + 20001 is still examined (:66)
    dx = x[lo:hi] - x[c]
    dy = y[lo:hi] - y[c]
    d2 = dx * dx + dy * dy
    # ring r (0-based) holds md[r]^2 < d2 <= md[r+1]^2
    ring = np.searchsorted(md * md, d2, side="left") - 1
    out = []
    for lev in range(levels):
        idx = np.flatnonzero(ring == lev) + lo
        if idx.size == 0:
            raise RuntimeError("Got no wells for cluster %s at (%s,%s) level %s"
                               % (c, int(x[c]), int(y[c]), lev))
        out.append(idx)
    return out


def generate(x: np.ndarray, y: np.ndarray, centres: Sequence[int], levels: int = 5,
             max_dists: Sequence[int] | None = None):
    """All targets: returns [(centre, [ring arrays])] in the given order."""
    seen = set()
    res = []
    for c in centres:
        assert c not in seen                                   # :159
        seen.add(c)
        res.append((int(c), rings_for(int(c), x, y, levels, max_dists)))
    return res


def write_targets(targets, fh) -> None:
    """The targets-file text (:162-167)."""
    for c, rings in targets:
        fh.write("%d\n" % c)
        for r in rings:
            fh.write(",".join(map(str, r.tolist())) + "\n")


def main(argv=None) -> int:
    """CLI with the reference's flags (:90-95) plus --levels."""
    from argparse import ArgumentParser, ArgumentDefaultsHelpFormatter
    p = ArgumentParser(description="Pick n random clusters from an s.locs file and list the "
                       "indexes of the surrounding wells, level by level.",
                       formatter_class=ArgumentDefaultsHelpFormatter)
    p.add_argument("-f", "--slocs", dest="slocs", type=str, required=True,
                   help="The slocs file to analyse.")
    p.add_argument("-s", "--seed", dest="seed", type=int, default=None,
                   help="Seed for the random read selection")
    p.add_argument("-n", "--sample_size", dest="sample_size", type=int, default=DEF_SAMPLE_SIZE,
                   help="number of n random clusters")
    p.add_argument("--levels", type=int, default=5,
                   help="number of levels to emit (the reference is fixed at 5)")
    args = p.parse_args(argv)
    log = lambda m: print(str(m), file=sys.stderr)
    log("seed: %s" % (args.seed))
    log("sample size: %s" % (args.sample_size))
    x, y = read_slocs(args.slocs)
    log("Maximum number of cluster according to s.locs: %s" % x.shape[0])
    centres = sample_centres(x.shape[0], args.sample_size, args.seed)
    write_targets(generate(x, y, centres, args.levels), sys.stdout)
    return 0


if __name__ == "__main__":
    sys.exit(main())
