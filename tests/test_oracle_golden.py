"""Pin the CPU oracle (oracle/welldup_oracle.c + oracle.py) against outputs of the
unmodified reference (tests/golden/*.json, made by tools/make_golden.py).

For every golden run and every (lane, tile): regenerate the synthetic bytes of the wells
the targets touch, run the oracle, and require
  * the reference's lane_dupl (per-target (dups, length) for valid centres) exactly,
  * the reference's stderr duplicate log exactly (order included),
  * the reference's stdout exactly (through the oracle's report restatement),
for equality, Hamming <= k and Levenshtein <= k, single and multi-range --cycles.
"""
import numpy as np
import pytest

from helpers import (FIXTURES, MODE_ID, compact_tile, expected_dup_log, fixture_targets,
                     lane_dupl_from, load_fixture, run_cycles)
from oracle import oracle
from well_duplicates_amd import synth


@pytest.mark.parametrize("name", FIXTURES)
def test_oracle_reproduces_reference(name):
    fx = load_fixture(name)
    spec = synth.spec_from_dict(fx["spec"])
    targets, (centre, lvl_off, nbr) = fixture_targets(name)
    assert len(targets) == min(fx["n_targets"], len(targets))
    for run in fx["runs"]:
        assert run["exception"] is None
        cycles = run_cycles(run)
        mode, k = MODE_ID[run["mode"]], run["k"]
        log = []
        text = ""
        for lane_rec in run["lanes"]:
            lane = int(lane_rec["lane"])
            got = {}
            for tile in fx["tiles"]:
                planes, filt, c2, n2, _ = compact_tile(spec, lane, tile, cycles, centre, nbr,
                                                       excluded_cbcl=bool(fx.get("cbcl")))
                valid, dups, lens, dist = oracle.count_tile(planes, filt, c2, lvl_off, n2,
                                                            mode, k, want_dist=True)
                got[tile] = lane_dupl_from(valid, dups, lens)
                log += expected_dup_log(planes, c2, lvl_off, n2, centre, nbr, valid, dist,
                                        0 if run["mode"] == "eq" else k)
            assert got == lane_rec["lane_dupl"], (name, run["flags"], lane)
            assert lane_rec["sample_size"] == len(targets)
            ld = {t: [[tuple(x) for x in targ] for targ in v] for t, v in got.items()}
            text += oracle.py_output_writer(lane_rec["lane"], len(targets), ld,
                                            verbose="-S" not in run["flags"])
        assert text == run["stdout"]
        if "-q" not in run["flags"]:
            assert log == run["dup_log"], (name, run["flags"])


def test_python_restatement_matches_c_oracle():
    """The pure-Python (reference-structured) restatement and the C oracle agree."""
    fx = load_fixture("mid")
    spec = synth.spec_from_dict(fx["spec"])
    targets, (centre, lvl_off, nbr) = fixture_targets("mid")
    coords = [t.coords for t in targets]
    cycles = list(range(0, 30))
    planes, filt, c2, n2, wells = compact_tile(spec, 1, "1101", cycles, centre, nbr)
    remap = {int(w): i for i, w in enumerate(wells)}
    coords_c = [[[remap[w] for w in ring] for ring in c] for c in coords]
    seqs = oracle.py_get_seqs([p.tobytes() for p in planes], filt.tobytes(),
                              [w for c in coords_c for ring in c for w in ring])
    for mode, metric, k in ((1, oracle.py_hamming, 1), (2, oracle.py_levenshtein, 2),
                            (0, oracle.py_hamming, 0)):
        want = oracle.py_count_tile(coords_c, seqs, fx["levels"], metric, k)
        valid, dups, lens, _ = oracle.count_tile(planes, filt, c2, lvl_off, n2, mode, k)
        got = [[tuple(x) for x in t] for t in lane_dupl_from(valid, dups, lens)]
        assert got == want


def test_distance_functions():
    rng = np.random.default_rng(3)
    assert oracle.levenshtein("ACGT", "ACGT") == 0
    assert oracle.levenshtein("ACGTN", "CGTNA") == 2
    assert oracle.hamming("ACGTN", "CGTNA") == 5
    assert oracle.levenshtein("kitten", "sitting") == 3
    for _ in range(300):
        n = int(rng.integers(0, 30))
        a = "".join(rng.choice(list("ACGTN"), n))
        b = list(a)
        for _ in range(int(rng.integers(0, 4))):
            op = rng.integers(0, 3)
            pos = int(rng.integers(0, len(b) + 1))
            if op == 0 and b:
                b[min(pos, len(b) - 1)] = str(rng.choice(list("ACGTN")))
            elif op == 1:
                b.insert(pos, str(rng.choice(list("ACGTN"))))
            elif b:
                del b[min(pos, len(b) - 1)]
        b = "".join(b)
        assert oracle.levenshtein(a, b) == oracle.py_levenshtein(a, b)
        if len(a) == len(b):
            assert oracle.hamming(a, b) == oracle.py_hamming(a, b)
    with pytest.raises(ValueError):
        oracle.hamming("AC", "A")


# Known answers for the third-party metric (count_well_duplicates.py:9, :200, :252).  The reference
# pins nothing here; these are the examples the python-Levenshtein package publishes in the
# docstrings of `distance` and `hamming` (7 and 5 for both functions), plus the editops example's
# pair ('spam' -> 'park' takes three operations).
LEVENSHTEIN_DOC_EXAMPLES = [("Hello world!", "Holly grail!", 7, 7), ("Brian", "Jesus", 5, 5), ("spam", "park", 3, 4)]
# Pairs over the reads' own alphabet with answers worked out by hand from the two definitions:
# (a, b, edit distance, Hamming distance) - the GPU test sends the same table through the HIP path.
DNA_KNOWN_ANSWERS = [
    ("ACGTACGTAC", "ACGTACGTAC", 0, 0),
    ("ACGTACGTAC", "ACGTTCGTAC", 1, 1),            # one substitution
    ("ACGTACGTAC", "CGTACGTACA", 2, 10),           # shifted by one: delete the first base, append one
    ("ACGTACGTAC", "AACGTACGTA", 2, 9),            # shifted the other way: only the first base agrees in place
    ("ACGTNACGTN", "ACGTAACGTN", 1, 1),            # a no-call is a symbol of its own: N vs A mismatches
    ("NNNNNNNNNN", "NNNNNNNNNN", 0, 0),            # ... and N matches N (bcl_direct_reader.py:181)
    ("AAAAACCCCC", "CCCCCAAAAA", 10, 10),
    ("ACACACACAC", "CACACACACA", 2, 10),
    ("ACGTACGTAC", "ACGTCAGTAC", 2, 2),            # a transposition costs two
    ("AGGTCACTGA", "AGTCACTGAA", 2, 7),            # delete the third base, append one: AG.. and ..A agree in place
]


def test_published_and_hand_checked_distances():
    for a, b, lev, ham in LEVENSHTEIN_DOC_EXAMPLES:
        assert oracle.py_levenshtein(a, b) == lev
        if len(a) == len(b):
            assert oracle.py_hamming(a, b) == ham
    for a, b, lev, ham in DNA_KNOWN_ANSWERS:
        assert oracle.py_levenshtein(a, b) == lev == oracle.levenshtein(a, b), (a, b)
        assert oracle.py_hamming(a, b) == ham == oracle.hamming(a, b), (a, b)


def test_oracle_errors():
    planes = [np.array([1, 2, 3, 4], dtype=np.uint8)]
    filt = np.ones(4, dtype=np.uint8)
    centre = np.array([0], dtype=np.int32)
    with pytest.raises(IndexError):      # bcl_direct_reader.py:186
        oracle.count_tile(planes, filt, centre, np.array([[0, 1]], np.int32),
                          np.array([4], np.int32), 0, 0)
    with pytest.raises(IndexError):      # :191 negative
        oracle.count_tile(planes, filt, centre, np.array([[0, 1]], np.int32),
                          np.array([-1], np.int32), 0, 0)
    with pytest.raises(AssertionError):  # count_well_duplicates.py:249
        oracle.count_tile(planes, filt, centre, np.array([[0, 1, 1]], np.int32),
                          np.array([1], np.int32), 0, 0)
    # an invalid centre is skipped before the empty-level assert can fire (:236-237)
    filt0 = np.zeros(4, dtype=np.uint8)
    valid, dups, lens, _ = oracle.count_tile(planes, filt0, centre, np.array([[0, 1, 1]], np.int32),
                                             np.array([1], np.int32), 0, 0)
    assert valid.tolist() == [0] and dups.tolist() == [[-1, -1]]
