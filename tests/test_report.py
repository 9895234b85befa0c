"""Tally arithmetic + report text.

Known answers: the reference's hand-built LANE_DUPL tables
(test/test_count_well_duplicates.py:37-91) and extra cases, each run through the reference's
output_writer as it is today (tests/golden/report_tables.json, made by tools/make_golden.py;
the 4 trailer lines are part of it - SURVEY.md F6).  Checked against
  * the product printer (well_duplicates_amd.report), from per-target stats and from the
    device-style histogram block,
  * the oracle's restatement (oracle.py_output_writer, oracle C tally).
"""
import io
import json
import os

import numpy as np
import pytest

from helpers import GOLD, FIXTURES, load_fixture
from oracle import oracle
from well_duplicates_amd import report

CASES = json.load(open(os.path.join(GOLD, "report_tables.json")))


def as_tuples(lane_dupl):
    return {t: [[tuple(x) for x in targ] for targ in v] for t, v in lane_dupl.items()}


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_report_tables(case):
    ld = as_tuples(case["lane_dupl"])
    kw = dict(case["kwargs"])
    want = case["stdout"]
    assert oracle.py_output_writer(case["lane"], case["sample_size"], ld, **kw) == want
    buf = io.StringIO()
    report.output_writer(case["lane"], case["sample_size"], ld, out=buf, **kw)
    assert buf.getvalue() == want


def test_reference_expected_tables_prefix():
    """The literal EXPECTED_OUT_1 of the reference's test file (per-level lines) is a prefix
    of what the code prints today (F6): whitespace runs -> tabs as its _rescmp does."""
    expected = """
Lane: 1   Tile: 1208   Targets: 4/4
Level: 1   Wells: 24   Dups: 5   Hit: 2   AccO: 2   AccI: 3
Level: 2   Wells: 44   Dups: 3   Hit: 3   AccO: 3   AccI: 3
Level: 3   Wells: 60   Dups: 1   Hit: 1   AccO: 3   AccI: 1
LaneSummary: 1   Tiles: 1   Targets: 4/4
Level: 1   Wells: 24   Dups: 5 (0.20833)   Hit: 2 (0.50000)   AccO: 2 (0.50000)   AccI: 3 (0.75000)
Level: 2   Wells: 44   Dups: 3 (0.06818)   Hit: 3 (0.75000)   AccO: 3 (0.75000)   AccI: 3 (0.75000)
Level: 3   Wells: 60   Dups: 1 (0.01667)   Hit: 1 (0.25000)   AccO: 3 (0.75000)   AccI: 1 (0.25000)
"""
    import re
    lines = [re.sub(r"\s\s+", "\t", s) for s in expected.strip().split("\n")]
    case = [c for c in CASES if c["name"] == "full"][0]
    buf = io.StringIO()
    report.output_writer(1, 4, as_tuples(case["lane_dupl"]), verbose=1, out=buf)
    got = buf.getvalue().rstrip("\n").split("\n")
    assert got[:len(lines)] == lines
    assert got[len(lines):] == ["", "Overall duplication (Acc/Targets): 75.00%",
                                "Picard-equivalent duplication v1:  56.25%",
                                "Picard-equivalent duplication v2:  62.50%"]


def test_histogram_block_equals_per_target_reduction():
    """first/last histograms -> AccO/AccI == the reference's explicit loops, random stats."""
    rng = np.random.default_rng(7)
    for levels in (1, 2, 3, 5, 7):
        for _ in range(20):
            T = int(rng.integers(0, 40))
            dups = rng.integers(0, 3, size=(T, levels)) * (rng.random((T, levels)) < 0.3)
            lens = rng.integers(1, 30, size=(T, levels))
            valid = (rng.random(T) < 0.7).astype(np.uint8)
            block = oracle.tally_tile(valid, dups.astype(np.int32), lens.astype(np.int32))
            stats = [[(int(dups[t, l]), int(lens[t, l])) for l in range(levels)]
                     for t in range(T) if valid[t]]
            tc = report.TileCounts.from_target_stats(stats, levels)
            assert tc.targets == block[0]
            assert tc.wells == block[1:1 + levels].tolist()
            assert tc.dups == block[1 + levels:1 + 2 * levels].tolist()
            assert tc.hit == block[1 + 2 * levels:1 + 3 * levels].tolist()
            assert tc.acco() == block[1 + 3 * levels:1 + 4 * levels].tolist()
            assert tc.acci() == block[1 + 4 * levels:1 + 5 * levels].tolist()
            # round trip through the device block layout
            dev = [tc.targets] + tc.wells + tc.dups + tc.hit + tc.first + tc.last
            tc2 = report.TileCounts.from_block(dev, levels)
            assert tc2 == tc


@pytest.mark.parametrize("name", FIXTURES)
def test_golden_stdout_from_lane_dupl(name):
    """Reference stdout of every golden run, reproduced from its own lane_dupl."""
    fx = load_fixture(name)
    for run in fx["runs"]:
        verbose = "-S" not in run["flags"]
        text_o, text_p = "", io.StringIO()
        for lane in run["lanes"]:
            ld = as_tuples(lane["lane_dupl"])
            text_o += oracle.py_output_writer(lane["lane"], lane["sample_size"], ld, verbose=verbose)
            report.output_writer(lane["lane"], lane["sample_size"], ld, verbose=verbose, out=text_p)
        assert text_o == run["stdout"]
        assert text_p.getvalue() == run["stdout"]


def test_zero_duplicates_divergence():
    """the reference raises ZeroDivisionError when a lane has valid targets
    but no duplicate; the product prints 0.00 % (strict=True reproduces the exception)."""
    ld = {"1101": [[(0, 6), (0, 12)], [(0, 6), (0, 11)]]}
    with pytest.raises(ZeroDivisionError):
        oracle.py_output_writer(1, 2, ld, verbose=1)
    with pytest.raises(ZeroDivisionError):
        report.output_writer(1, 2, ld, verbose=1, out=io.StringIO(), strict=True)
    buf = io.StringIO()
    report.output_writer(1, 2, ld, verbose=1, out=buf)
    assert buf.getvalue().endswith(
        "\nOverall duplication (Acc/Targets): 0.00%\n"
        "Picard-equivalent duplication v1:  0.00%\n"
        "Picard-equivalent duplication v2:  0.00%\n")
    assert "Level: 2\tWells: 23\tDups: 0 (0.00000)\tHit: 0 (0.00000)" in buf.getvalue()
