"""Tally reducer and report printer for the scan path.

Host-side mirror of `output_writer` (count_well_duplicates.py:27-153).  The device
returns, per tile, the integer block

    [valid, wells[0..n), dups[0..n), hit[0..n), first[0..n), last[0..n)]      (n = levels)

where for every valid target with per-level dup counts d[l] and hit mask m = {l : d[l] > 0}
    wells[l] += len(ring l)     dups[l] += d[l]     hit[l] += (l in m)
    first[min(m)] += 1          last[max(m)] += 1                (nothing if m is empty)
and the reference's accumulated columns are prefix / suffix sums of those histograms:
    AccO[l] = sum(first[0..l])  (count_well_duplicates.py:80-84, inside-out)
    AccI[l] = sum(last[l..n))   (count_well_duplicates.py:85-89, outside-in)
All of it is integer arithmetic; the only floating point is the printed ratios, which stay
on the host as Python floats evaluated in the reference's order (:115-123, :135-153).

Known divergence: with valid targets but zero duplicates in a lane the
reference dies with ZeroDivisionError at :115-117 after printing the per-tile lines.
This printer reports 0.00 % instead; pass strict=True to get the reference's exception.
"""
from __future__ import annotations

import sys
from dataclasses import dataclass, field
from typing import Dict, List, Sequence

TALLY = 0    # count_well_duplicates.py:17
LENGTH = 1   # count_well_duplicates.py:18


@dataclass
class TileCounts:
    """Per-tile integer tallies for `levels` levels."""
    targets: int = 0
    wells: List[int] = field(default_factory=list)
    dups: List[int] = field(default_factory=list)
    hit: List[int] = field(default_factory=list)
    first: List[int] = field(default_factory=list)
    last: List[int] = field(default_factory=list)

    @property
    def levels(self) -> int:
        return len(self.wells)

    @classmethod
    def zeros(cls, levels: int) -> "TileCounts":
        z = lambda: [0] * levels
        return cls(0, z(), z(), z(), z(), z())

    @classmethod
    def from_block(cls, block: Sequence[int], levels: int) -> "TileCounts":
        """Decode one row of the device counter block (include/welldup.h, wd_count_tiles)."""
        b = [int(v) for v in block]
        assert len(b) == 1 + 5 * levels
        cut = lambda k: b[1 + k * levels: 1 + (k + 1) * levels]
        return cls(b[0], cut(0), cut(1), cut(2), cut(3), cut(4))

    @classmethod
    def from_target_stats(cls, tile_counts, levels: int) -> "TileCounts":
        """Reduce the reference's per-target list [[(tally, length)] * levels] * targets."""
        c = cls.zeros(levels)
        c.targets = len(tile_counts)
        for targ in tile_counts:
            mask = [bool(targ[lev][TALLY]) for lev in range(levels)]
            for lev in range(levels):
                c.wells[lev] += targ[lev][LENGTH]
                c.dups[lev] += targ[lev][TALLY]
                c.hit[lev] += mask[lev]
            if any(mask):
                c.first[mask.index(True)] += 1
                c.last[levels - 1 - mask[::-1].index(True)] += 1
        return c

    def acco(self) -> List[int]:
        out, run = [], 0
        for v in self.first:
            run += v
            out.append(run)
        return out

    def acci(self) -> List[int]:
        out, run = [], 0
        for v in reversed(self.last):
            run += v
            out.append(run)
        return out[::-1]


def write_report(lane, sample_size, tiles: Dict[str, TileCounts], levels: int = 0,
                 verbose: bool = False, out=None, strict: bool = False) -> None:
    """Print the per-tile (verbose) and per-lane report exactly as the reference does.

    tiles: {tile id string: TileCounts}; printed in sorted string order (:63).
    levels: 0 = infer from the first tile with a valid target (:41-47); if no tile has
            one, no per-level line is printed at all.
    """
    out = out or sys.stdout
    if not levels:
        for tc in tiles.values():
            if tc.targets > 0:
                levels = tc.levels
                break

    tot_targets = 0
    tot = TileCounts.zeros(levels)
    tot_acco = [0] * levels
    tot_acci = [0] * levels

    for tile in sorted(tiles.keys()):
        tc = tiles[tile]
        tot_targets += tc.targets
        if verbose:
            print("Lane: %s\tTile: %s\tTargets: %i/%i" % (lane, tile, tc.targets, sample_size),
                  file=out)
        if tc.levels != levels:
            # Only a tile with no valid target may lack level data.  Truncating a
            # histogram to fewer levels is not possible (AccI needs per-target data):
            # callers reduce with the wanted level count instead.
            if tc.targets != 0:
                raise ValueError("tile %s holds %d levels, report wants %d"
                                 % (tile, tc.levels, levels))
            tc = TileCounts.zeros(levels)
        acco, acci = tc.acco(), tc.acci()
        for lev in range(levels):
            if verbose:
                print("Level: %i\tWells: %i\tDups: %i\tHit: %i\tAccO: %i\tAccI: %i" % (
                    lev + 1, tc.wells[lev], tc.dups[lev], tc.hit[lev], acco[lev], acci[lev]),
                    file=out)
            tot.wells[lev] += tc.wells[lev]
            tot.dups[lev] += tc.dups[lev]
            tot.hit[lev] += tc.hit[lev]
            tot_acco[lev] += acco[lev]
            tot_acci[lev] += acci[lev]

    # "Picard-equivalent" percentages, operation order as :111-125
    if tot_acci:
        grand_tot_hits = tot_acci[0]
        grand_tot_dups = sum(tot.dups)
        if strict or (grand_tot_dups + grand_tot_hits) != 0:
            peds = (grand_tot_hits *
                    (1 - grand_tot_hits / (grand_tot_dups + grand_tot_hits)) /
                    tot_targets)
            peds2 = (grand_tot_hits *
                     (1 - grand_tot_hits / (2 * grand_tot_dups)) /
                     tot_targets)
        else:
            peds = peds2 = 0          # reference: ZeroDivisionError
    else:
        grand_tot_hits = peds = peds2 = 0

    print("LaneSummary: %s\tTiles: %i\tTargets: %i/%i" % (
        lane, len(tiles), tot_targets, sample_size * len(tiles)), file=out)

    for lev in range(levels):
        if strict or tot_targets:
            r_dups = tot.dups[lev] / tot.wells[lev]
            r_hit = tot.hit[lev] / tot_targets
            r_acco = tot_acco[lev] / tot_targets
            r_acci = tot_acci[lev] / tot_targets
        else:
            r_dups = r_hit = r_acco = r_acci = 0.0
        print("Level: %i\tWells: %i\tDups: %i (%.5f)\t" % (
            lev + 1, tot.wells[lev], tot.dups[lev], r_dups) +
            "Hit: %i (%.5f)\tAccO: %i (%.5f)\tAccI: %i (%.5f)" % (
                tot.hit[lev], r_hit, tot_acco[lev], r_acco, tot_acci[lev], r_acci),
            file=out)

    raw_dup_rate = grand_tot_hits / tot_targets if grand_tot_hits else 0.0

    print(file=out)
    print("Overall duplication (Acc/Targets): {:.2%}".format(raw_dup_rate), file=out)
    print("Picard-equivalent duplication v1:  {:.2%}".format(peds), file=out)
    print("Picard-equivalent duplication v2:  {:.2%}".format(peds2), file=out)


def output_writer(lane, sample_size, lane_dupl, levels=0, verbose=False, out=None,
                  strict=False):
    """Reference signature (count_well_duplicates.py:27): report from per-target stats.

    lane_dupl: {tile: [[(tally, length)] * levels] * valid_targets}.  With `levels` given,
    only the first `levels` entries of each target are reduced, as the reference does.
    """
    if not levels:
        for atile in lane_dupl.values():
            if len(atile) > 0:
                levels = len(atile[0])
                break
    tiles = {tile: TileCounts.from_target_stats(tc, levels) for tile, tc in lane_dupl.items()}
    write_report(lane, sample_size, tiles, levels=levels, verbose=verbose, out=out,
                 strict=strict)
