"""Targets-file model: host-side mirror of the reference's target.py for the scan path.

Same names, argument meaning and error behaviour as the reference
(target.py:6-40 `load_targets`, :42-106 `AllTargets`, :108-144 `Target`), so the
reference's own test_target.py assertions apply unchanged (tests/test_targets.py).
What is new is `AllTargets.to_csr()`: the ragged per-level neighbour lists flattened
into the CSR arrays the device keeps resident (level lists are ragged,
never a dense [T, 6*level] tensor).

File format: a record is one line holding a single integer
(the centre well, no comma) followed by one comma-separated line per level.  A line
without a comma starts the next record.
"""
from __future__ import annotations

import re

import numpy as np


_REGULAR = re.compile(r"(?:-?[0-9]+[,\n])*(?:-?[0-9]+)?")


def _regular(text: str) -> bool:
    """`_REGULAR.fullmatch(text)` - decimal integers, each followed by one comma or newline (the last
    one may go without) - in a few numpy passes over the bytes: the regular expression costs 20-50 ms
    on the 1.7 MB of a 2500-target file, a tenth of a lane's run."""
    try:
        b = np.frombuffer(text.encode("ascii"), dtype=np.uint8)
    except UnicodeEncodeError:
        return False
    if b.size == 0:
        return True
    digit = (b >= 48) & (b <= 57)
    sep = (b == 44) | (b == 10)
    minus = b == 45
    if not (digit | sep | minus).all():
        return False
    if sep[0] or minus[-1]:
        return False
    # a separator follows a digit; a minus follows a separator (or opens the text) and precedes a digit
    if (sep[1:] & ~digit[:-1]).any():
        return False
    if (minus[1:] & ~sep[:-1]).any() or (minus[:-1] & ~digit[1:]).any():
        return False
    return True


class Target:
    """One centre well plus its per-level neighbour wells: [[c], [lvl 1 ...], [lvl 2 ...]]."""

    __slots__ = ("coords",)

    def __init__(self, coords):
        # target.py:113 - the centre line must hold exactly one integer
        assert len(coords[0]) == 1, "Centre of target must be a single int, not " + str(coords)
        self.coords = coords

    def get_indices(self, level=None):
        if level is None:
            return [w for ring in self.coords for w in ring]
        return self.coords[level]

    def get_centre(self):
        return self.coords[0][0]

    def get_levels(self):
        """Number of lines in the record, the centre line included."""
        return len(self.coords)

    def get_level_from_index(self, index):
        for lev, ring in enumerate(self.coords):
            if index in ring:
                return lev
        return None


class AllTargets:
    """Ordered collection of Target objects (file order == iteration order, target.py:59-61)."""

    def __init__(self):
        self._by_centre = {}
        self._wells = None        # well index -> [Target, ...] in insertion order; built on first use
        self.levels = None

    def _reverse(self):
        """The reverse index of target.py:80-83, built when first asked for (the scan path never is:
        it works from to_csr(), and 215 000 dictionary updates are most of a targets file's parse)."""
        if self._wells is None:
            self._wells = {}
            for t in self._by_centre.values():
                for w in t.get_indices():
                    self._wells.setdefault(w, []).append(t)
        return self._wells

    def __len__(self):
        return len(self._by_centre)

    def __iter__(self):
        return iter(self._by_centre.values())

    def get_target_by_centre(self, centre):
        return self._by_centre[centre]

    def add_target(self, coords):
        t = Target(coords)
        # target.py:72 - a centre may appear once only
        assert t.get_centre() not in self._by_centre
        # target.py:75-78 - every record has the same number of lines
        if self.levels is None:
            self.levels = t.get_levels()
        else:
            assert self.levels == t.get_levels()
        self._by_centre[t.get_centre()] = t
        if self._wells is not None:
            for w in t.get_indices():
                self._wells.setdefault(w, []).append(t)

    def get_all_indices(self, level=None):
        """All well indices held; level=0 centres only, level=n that ring only (with repeats)."""
        if level == 0:
            return list(self._by_centre.keys())
        if level is None:
            return list(self._reverse().keys())
        return [w for t in self for w in t.get_indices(level)]

    def get_from_index(self, index):
        return [(t, t.get_level_from_index(index)) for t in self._reverse().get(index, [])]

    # ---------------------------------------------------------------- device layout
    def to_csr(self, levels=None):
        """Flatten rings 1..levels of every target, in file order.

        Returns (centre[T] int32, lvl_off[T, levels+1] int32, nbr[P] int32) where the
        neighbours of target t at level l (1-based, as `Target.get_indices(l)`) are
        nbr[lvl_off[t, l-1] : lvl_off[t, l]].  Offsets are absolute into nbr.
        """
        have = (self.levels or 1) - 1
        if levels is None:
            levels = have
        if levels > have:
            raise ValueError("targets file holds %d levels, %d requested" % (have, levels))
        T = len(self)
        centre = np.empty(T, dtype=np.int64)
        lvl_off = np.zeros((T, levels + 1), dtype=np.int64)
        flat = []
        pos = 0
        for i, t in enumerate(self):
            centre[i] = t.get_centre()
            lvl_off[i, 0] = pos
            for lev in range(1, levels + 1):
                ring = t.coords[lev]
                flat.extend(ring)
                pos += len(ring)
                lvl_off[i, lev] = pos
        nbr = np.asarray(flat, dtype=np.int64) if flat else np.zeros(0, dtype=np.int64)
        for name, arr in (("centre", centre), ("neighbour", nbr)):
            if arr.size and (arr.min() < -(2 ** 31) or arr.max() >= 2 ** 31):
                raise OverflowError("%s index does not fit int32" % name)
        if pos >= 2 ** 31:
            raise OverflowError("too many neighbour slots for int32 offsets")
        return (centre.astype(np.int32), lvl_off.astype(np.int32), nbr.astype(np.int32))


def load_targets(filename, levels=None, limit=None):
    """Parse a targets file.

    levels: number of lines per record to keep, the centre line included (so the CLI
            passes `-l` + 1, count_well_duplicates.py:202-204); None keeps all.
    limit:  stop after this many targets (falsy = no limit).

    Errors as the reference: a blank line raises ValueError (int('')); records of
    differing length or a repeated centre raise AssertionError.
    """
    all_targets = AllTargets()
    pending = None
    with open(filename, "r") as fh:
        lines = [ln.rstrip() for ln in fh]
    lines.append("")              # sentinel: flushes the last record (target.py:25)
    for ln in lines:
        if "," not in ln:
            if pending:
                all_targets.add_target([list(map(int, rec.split(","))) for rec in pending[:levels]])
                if limit and len(all_targets) == limit:
                    break
            pending = []
        pending.append(ln)
    return all_targets


def load_targets_csr(filename, level, limit=None):
    """The scan path's view of a targets file without the object model: (number of targets,
    centre[T], lvl_off[T, level+1], nbr[P]) exactly as `load_targets(filename, level + 1,
    limit).to_csr(level)` gives them, parsed in bulk (numpy) - or None whenever the file is not
    perfectly regular (a line that is not comma-separated integers, records of different lengths,
    fewer rings than asked for, a repeated centre, values beyond int32, trailing blanks): the caller
    then takes load_targets(), which raises what the reference raises (target.py:6-40, :72-78)."""
    with open(filename, "r") as fh:
        text = fh.read()
    if not text or not _regular(text):
        # anything but decimal integers separated by single commas / newlines ('-' alone, '0x10',
        # '7abc', blanks, '+5', '1_0' ...): numpy's bulk parser would read some of those as numbers
        # where int() raises (target.py:31) - the reference's parser decides
        return None
    # commas per line, from the bytes (a regular text has no empty line: every newline follows a digit)
    b = np.frombuffer(text.encode("ascii"), dtype=np.uint8)
    ends = np.flatnonzero(b == 10)
    if b[-1] == 44:
        return None                         # the text ends in a comma: an empty last token
    if b[-1] != 10:
        ends = np.append(ends, b.size)      # the last line has no newline of its own
    comma_before = np.concatenate([[0], np.cumsum(b == 44)])
    commas = np.diff(np.concatenate([[0], comma_before[ends]])).astype(np.int64)
    n_lines = int(ends.size)
    starts = np.flatnonzero(commas == 0)    # a line without a comma starts a record (target.py:27)
    if starts.size == 0 or starts[0] != 0:
        return None
    per = int(starts[1]) if starts.size > 1 else n_lines
    if per < level + 1 or n_lines % per or not np.array_equal(starts, np.arange(0, n_lines, per)):
        return None
    import warnings
    try:
        with warnings.catch_warnings():     # (numpy warns when it stops at something that is not a number)
            warnings.simplefilter("ignore")
            flat = np.fromstring(text.replace("\n", ","), dtype=np.int64, sep=",")
    except ValueError:
        return None
    if flat.size != int(commas.sum()) + n_lines:
        return None                         # something that is not an integer stopped the parse
    if flat.size and (flat.min() < -(2 ** 31) or flat.max() >= 2 ** 31):
        return None
    n_rec = n_lines // per
    T = min(n_rec, int(limit)) if limit else n_rec
    line_len = (commas + 1).reshape(n_rec, per)[:T]
    line_off = np.concatenate([[0], np.cumsum(commas + 1)])[:-1].reshape(n_rec, per)[:T]
    centre = flat[line_off[:, 0]]
    if np.unique(centre).size != T:
        return None                         # a centre twice: load_targets asserts (target.py:72)
    ring_len = line_len[:, 1:level + 1]
    lvl_off = np.zeros((T, level + 1), dtype=np.int64)
    lvl_off[:, 1:] = np.cumsum(ring_len, axis=1)
    lvl_off += np.concatenate([[0], np.cumsum(ring_len.sum(axis=1))[:-1]])[:, None]
    if lvl_off.size and lvl_off[-1, -1] >= 2 ** 31:
        return None
    # the kept lines of every record are contiguous in `flat`: lines 1..level of record t
    first = line_off[:, 1]
    total = ring_len.sum(axis=1)
    idx = np.repeat(first - np.concatenate([[0], np.cumsum(total)[:-1]]), total) + np.arange(int(total.sum()))
    nbr = flat[idx]
    return T, centre.astype(np.int32), lvl_off.astype(np.int32), nbr.astype(np.int32)
