"""Property tests (hypothesis) of the host logic: targets-file round trips and the tally
reducer against the oracle's restatement of the reference's loops."""
import io
import os

from hypothesis import given, settings, strategies as st

from oracle import oracle
from well_duplicates_amd import report
from well_duplicates_amd.targets import load_targets

ring = st.lists(st.integers(0, 5_000_000), min_size=2, max_size=12)     # a 1-element ring is ambiguous in the file format


@st.composite
def targets_files(draw):
    levels = draw(st.integers(1, 6))
    n = draw(st.integers(1, 12))
    centres = draw(st.lists(st.integers(0, 5_000_000), min_size=n, max_size=n, unique=True))
    return [[[c]] + [draw(ring) for _ in range(levels)] for c in centres]


@settings(max_examples=60, deadline=None)
@given(targets_files(), st.integers(0, 8), st.integers(0, 14))
def test_targets_file_roundtrip(tmp_path_factory, recs, cut_levels, limit):
    path = os.path.join(str(tmp_path_factory.mktemp("t")), "x.list")
    with open(path, "w") as fh:
        for rec in recs:
            for line in rec:
                fh.write(",".join(map(str, line)) + "\n")
    have = len(recs[0])
    keep = None if cut_levels == 0 else min(cut_levels, have)
    t = load_targets(path, levels=keep, limit=limit or None)
    want = oracle.py_load_targets(path, levels=keep, limit=limit or None)
    assert [x.coords for x in t] == want
    n = len(recs) if not limit else min(limit, len(recs))
    assert len(t) == n
    lv = (keep or have) - 1
    centre, lvl_off, nbr = t.to_csr(lv)
    for i, rec in enumerate(recs[:n]):
        assert centre[i] == rec[0][0]
        for l in range(1, lv + 1):
            assert nbr[lvl_off[i, l - 1]:lvl_off[i, l]].tolist() == rec[l]
    assert set(t.get_all_indices()) == {w for rec in recs[:n] for line in rec[:lv + 1] for w in line}


stats = st.lists(st.lists(st.tuples(st.integers(0, 4).map(lambda v: v if v < 3 else 0), st.integers(1, 40)),
                          min_size=4, max_size=4), min_size=0, max_size=15)


@settings(max_examples=80, deadline=None)
@given(st.dictionaries(st.sampled_from(["1101", "1102", "1203", "2228", "2101"]), stats, min_size=1, max_size=4),
       st.booleans(), st.integers(0, 4))
def test_report_equals_reference_restatement(lane_dupl, verbose, levels):
    """Same text (or the same ZeroDivisionError) as the restated output_writer."""
    try:
        want = oracle.py_output_writer("3", 7, lane_dupl, levels=levels, verbose=verbose)
    except ZeroDivisionError:
        want = None
    buf = io.StringIO()
    try:
        report.output_writer("3", 7, lane_dupl, levels=levels, verbose=verbose, out=buf, strict=True)
        got = buf.getvalue()
    except ZeroDivisionError:
        got = None
    assert got == want
    # the graceful (non-strict) printer never raises and agrees whenever the reference survives
    buf2 = io.StringIO()
    report.output_writer("3", 7, lane_dupl, levels=levels, verbose=verbose, out=buf2)
    if want is not None:
        assert buf2.getvalue() == want


def _lev2_equal_length(a: str, b: str) -> int:
    """The closed form the dense path uses for Levenshtein <= 2 on equal-length reads
    (csrc/scan_dense.inc: lev2_window / lev2_words), restated on strings: the distance if it is
    <= 2, else 3."""
    mism = [i for i in range(len(a)) if a[i] != b[i]]
    if len(mism) <= 2:
        return len(mism)
    p, t = mism[0], mism[-1]
    if all(b[i] == a[i + 1] for i in range(p, t)):            # deletion at p, insertion at t
        return 2
    if all(b[i] == a[i - 1] for i in range(p + 1, t + 1)):    # insertion at p, deletion at t
        return 2
    return 3


@settings(max_examples=3000, deadline=None)
@given(st.data())
def test_levenshtein_le2_closed_form(data):
    alphabet = data.draw(st.sampled_from(["AC", "ACGTN", "A", "ACG"]))
    n = data.draw(st.integers(1, 24))
    a = data.draw(st.text(alphabet=alphabet, min_size=n, max_size=n))
    kind = data.draw(st.integers(0, 2))
    b = list(a)
    if kind == 0:                                              # a few substitutions
        for _ in range(data.draw(st.integers(0, 3))):
            b[data.draw(st.integers(0, n - 1))] = data.draw(st.sampled_from(alphabet))
    elif kind == 1 and n >= 2:                                 # delete one, insert one (+ maybe a substitution)
        del b[data.draw(st.integers(0, n - 1))]
        b.insert(data.draw(st.integers(0, n - 1)), data.draw(st.sampled_from(alphabet)))
        if data.draw(st.booleans()):
            b[data.draw(st.integers(0, n - 1))] = data.draw(st.sampled_from(alphabet))
    else:
        b = list(data.draw(st.text(alphabet=alphabet, min_size=n, max_size=n)))
    b = "".join(b)
    assert _lev2_equal_length(a, b) == min(oracle.py_levenshtein(a, b), 3)
    # on a prefix the same test is a necessary condition (the signature filter relies on it)
    if oracle.py_levenshtein(a, b) <= 2:
        for cut in range(1, n):
            assert _lev2_equal_length(a[:cut], b[:cut]) <= 2
    # the screen of the dense compare stage (csrc/scan_dense.inc: lev2_screen, on the first 16
    # cycles, 2 bits each, a no-call reading as A) never rejects a pair within distance 2
    if oracle.py_levenshtein(a, b) <= 2:
        for cut in (min(n, 16), min(n, 7), 1):
            assert _lev2_screen(_screen_word(a[:cut]), _screen_word(b[:cut])) <= 1, (a[:cut], b[:cut])
            # ... and neither does the exact test k_dense_mark runs on the same cycles (lev2_window16:
            # the closed form on the reads as the screen words hold them, a no-call reading as A)
            assert _lev2_equal_length(a[:cut].replace("N", "A"), b[:cut].replace("N", "A")) <= 2


def _screen_word(s: str) -> int:
    """k_dense_sig's screen word: byte & 3 per cycle (A, C, G, T = 0..3; a no-call, byte 0, reads
    as A), 2 bits each; unused high fields are 0."""
    return sum(("ACGTN".index(ch) & 3 if ch != "N" else 0) << (2 * i) for i, ch in enumerate(s))


def _lev2_screen(a: int, b: int) -> int:
    """csrc/scan_dense.inc lev2_screen on 32-bit words."""
    ah, al = a >> 2, (a << 2) & 0xFFFFFFFF
    fold = lambda x: (x | (x >> 1))
    m0 = fold(a ^ b) & 0x55555555
    up = fold(ah ^ b) & m0
    dn = fold(al ^ b) & m0
    pop = lambda v: bin(v).count("1")
    return min((pop(m0) - 1) & 0xFFFFFFFF, pop(up), pop(dn))
