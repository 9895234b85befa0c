"""Host side of the CLI: tile lists per sequencer type, -t patterns, cycle ranges and argument
defaults, pinned to what count_well_duplicates.py:163-197 and :280-315 of the reference compute
(values worked out from those lines)."""
import pytest

from well_duplicates_amd import count_well_duplicates as cwd
from well_duplicates_amd import workload


def test_tile_lists():
    x = workload.tiles_for_stype("hiseq_x")
    assert len(x) == 96 and x[0] == "1101" and x[23] == "1124" and x[24] == "1201" and x[-1] == "2224"
    k = workload.tiles_for_stype("hiseq_4000")
    assert len(k) == 112 and k[27] == "1128" and k[-1] == "2228"
    # a number is the highest tile id: %100 tiles per swath, //100 -> surfaces * 10 + swaths
    nova = workload.tiles_for_stype("2678")
    assert len(nova) == 2 * 6 * 78 and nova[0] == "1101" and nova[77] == "1178" and nova[-1] == "2678"
    assert workload.tiles_for_stype("24") == x                # //100 == 0 -> swaths stay 22
    assert workload.tiles_for_stype("anything else") == x     # ValueError swallowed: 24 / 22


def test_tile_patterns():
    tiles = workload.tiles_for_stype("hiseq_4000")
    assert workload.filter_tiles(tiles, "1101", "hiseq_4000") == ["1101"]
    assert workload.filter_tiles(tiles, "2228,1101,1101", "hiseq_4000") == ["1101", "2228"]       # sorted set
    top = workload.filter_tiles(tiles, "1...", "hiseq_4000")
    assert len(top) == 56 and all(t[0] == "1" for t in top)
    even_top = workload.filter_tiles(tiles, "1..[02468]", "hiseq_4000")
    assert len(even_top) == 28 and all(int(t) % 2 == 0 for t in even_top)
    with pytest.raises(AssertionError):                       # anchored: a prefix is not a match
        workload.filter_tiles(tiles, "110", "hiseq_4000")
    with pytest.raises(AssertionError):
        workload.filter_tiles(tiles, "1101,9999", "hiseq_4000")


def test_cycle_ranges():
    assert workload.parse_cycles(50, 100, None) == [(50, 100)]
    assert workload.parse_cycles(50, 100, "10-50,100-120") == [(10, 50), (100, 120)]
    assert workload.parse_cycles(0, 0, "0-1") == [(0, 1)]
    with pytest.raises(ValueError):
        workload.parse_cycles(0, 0, "10")                     # the reference's cryptic failure too
    with pytest.raises(ValueError):
        workload.parse_cycles(0, 0, "a-b")


def test_argument_defaults_and_flags():
    a = cwd.parse_args(["-f", "t.list", "-s", "hiseq_x", "-r", "/run"])
    assert (a.edit_distance, a.sample_size, a.level, a.start, a.end) == (2, 2500, 3, 50, 100)
    assert a.lane is None and a.tile_id is None and a.cycles is None
    assert not a.hamming and not a.summary_only and not a.quiet
    a = cwd.parse_args(["--coord_file", "t.list", "--stype", "2678", "--run", "/r", "--edit_distance", "0",
                        "--sample_size", "10", "--level", "5", "--tile", "1101,12..", "--lane", "3,4",
                        "--start", "0", "--end", "25", "--cycles", "0-10,20-30", "--hamming",
                        "--summary-only", "--quiet"])
    assert (a.coord_file, a.stype, a.run) == ("t.list", "2678", "/r")
    assert (a.edit_distance, a.sample_size, a.level, a.start, a.end) == (0, 10, 5, 0, 25)
    assert (a.tile_id, a.lane, a.cycles) == ("1101,12..", "3,4", "0-10,20-30")
    assert a.hamming and a.summary_only and a.quiet
    for missing in (["-s", "hiseq_x", "-r", "/r"], ["-f", "x", "-r", "/r"], ["-f", "x", "-s", "hiseq_x"]):
        with pytest.raises(SystemExit):
            cwd.parse_args(missing)
    # this package's extra: every well a centre, rings from s.locs - then no targets file
    a = cwd.parse_args(["--all-wells", "-s", "hiseq_x", "-r", "/r"])
    assert a.all_wells and a.coord_file is None and a.slocs is None


def test_mode_selection():
    # -e 0 is string equality whatever the metric; --hamming picks the positional count
    from well_duplicates_amd.scanner import MODE_EQ, MODE_HAMMING, MODE_LEVENSHTEIN, compare_mode
    assert compare_mode(0, False) == (MODE_EQ, 0) and compare_mode(0, True) == (MODE_EQ, 0)
    assert compare_mode(2, True) == (MODE_HAMMING, 2)
    assert compare_mode(2, False) == (MODE_LEVENSHTEIN, 2)


def test_resident_layout_choice():
    """--layout auto: the interleaved-by-four layout wherever the kernels that read it serve the run - sampled
    targets of at most 508 neighbour slots, equality / Hamming <= 254 / Levenshtein <= 3 (the reference's default
    is -e 2, count_well_duplicates.py:282-283), fewer than 65 536 targets - and planes for everything else."""
    import numpy as np
    from well_duplicates_amd.scanner import compare_mode

    def csr(T, per_target):
        off = (np.arange(T)[:, None] * per_target + np.array([0, per_target // 2, per_target])[None, :]).astype(np.int32)
        return np.arange(T, dtype=np.int32), off, np.zeros(T * per_target, dtype=np.int32)

    def layout(argv, c, cycles=(0, 1, 2, 3)):
        args = cwd.parse_args(["-s", "hiseq_x", "-r", "/nowhere"] + argv)
        mode, k = compare_mode(args.edit_distance, args.hamming)
        return cwd.resident_layout(args, mode, k, c, None, [1], ["1101"], list(cycles))

    small = csr(100, 86)
    assert layout(["-f", "x"], small) == 4                                   # the reference's defaults: -e 2, Levenshtein
    assert layout(["-f", "x", "-e", "0"], small) == 4
    assert layout(["-f", "x", "-e", "3"], small) == 4
    assert layout(["-f", "x", "-e", "4"], small) == 1                        # wider bands read planes
    assert layout(["-f", "x", "-e", "200", "--hamming"], small) == 4
    assert layout(["-f", "x", "-e", "255", "--hamming"], small) == 1         # beyond the 8-bit mismatch count
    assert layout(["-f", "x"], csr(100, 600)) == 1                           # more slots than four passes hold
    assert layout(["-f", "x"], csr(70000, 4)) == 1                           # 65 536 targets: the dense path, on planes
    assert layout(["--all-wells"], None) == 1
    assert layout(["-f", "x", "--layout", "planes"], small) == 1
    assert layout(["-f", "x", "--layout", "interleaved"], small) == 4
    with pytest.raises(SystemExit):                                          # the explicit layout is checked at once
        cwd.parse_args(["-s", "hiseq_x", "-r", "/nowhere", "-f", "x", "-e", "5", "--layout", "interleaved"])
