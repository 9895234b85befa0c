"""Run-directory reader (well_duplicates_amd.bcl): planes, filters, reference-style get_seqs,
error behaviour, and the CBCL (NovaSeq) path with and without excluded wells."""
import gzip
import os
import struct

import numpy as np
import pytest

from oracle import oracle
from well_duplicates_amd import bcl, synth

SPEC = synth.SynthSpec(seed=6, n_clusters=3001, row=50, nocall_per_64k=3000, plant_per_64k=9000)


@pytest.fixture(scope="module")
def run_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("run")
    synth.write_run_dir(SPEC, str(d), [1, 2], ["1101", "2205"], list(range(12)))
    return str(d)


def test_planes_and_filter_roundtrip(run_dir):
    rd = bcl.BCLReader(run_dir)
    assert sorted(rd.lanes) == ["L001", "L002"]
    for lane in (1, "2"):
        for tile in ("1101", 2205):
            t = rd.get_tile(lane, tile)
            assert t.num_clusters == SPEC.n_clusters and t.num_cycles == 12
            assert (t.read_filter() == synth.filter_bytes(SPEC, int(lane), int(tile))).all()
            for c in (0, 5, 11):
                assert (t.read_plane(c) == synth.plane_bytes(SPEC, int(lane), int(tile), c)).all()


def test_get_seqs_matches_reference_semantics(run_dir):
    t = bcl.BCLReader(run_dir).get_tile(1, "1101")
    idx = [0, 5, 17, 17, 3000, 1234]
    got = t.get_seqs(idx, 2, 9)
    planes = [synth.plane_bytes(SPEC, 1, 1101, c).tobytes() for c in range(2, 9)]
    want = oracle.py_get_seqs(planes, synth.filter_bytes(SPEC, 1, 1101).tobytes(), idx)
    assert got == want
    assert set("".join(s for s, _ in got.values())) <= set("ACGTN")
    assert len(t.get_seqs([1, 2])[1][0]) == 12          # end=None -> all cycle dirs
    with pytest.raises(IndexError):
        t.get_seqs([3001])
    with pytest.raises(IndexError):
        t.get_seqs([-1])


def test_errors(run_dir, tmp_path):
    rd = bcl.BCLReader(run_dir)
    with pytest.raises(RuntimeError):                     # no .filter for this tile (:131-132)
        rd.get_tile(1, "1199")
    t = rd.get_tile(1, "1101")
    with pytest.raises(FileNotFoundError):                # no C13.1 directory at all
        t.read_plane(12)
    with pytest.raises(FileNotFoundError):
        bcl.BCLReader(str(tmp_path))
    # plane whose header disagrees with the filter: AssertionError (:338)
    bad = os.path.join(run_dir, "Data", "Intensities", "BaseCalls", "L001", "C1.1", "s_1_1101.bcl.gz")
    keep = open(bad, "rb").read()
    try:
        with gzip.open(bad, "wb") as fh:
            fh.write(struct.pack("<I", 7) + b"\1" * 7)
        with pytest.raises(AssertionError):
            rd.get_tile(1, "1101").read_plane(0)
    finally:
        open(bad, "wb").write(keep)


@pytest.mark.parametrize("excluded", [False, True])
def test_cbcl(tmp_path, excluded):
    """NovaSeq layout: one .cbcl per (cycle, lane, surface) holding many tiles."""
    n = 2001                                              # odd: last byte half used
    spec = synth.SynthSpec(seed=9, n_clusters=n, row=40, nocall_per_64k=4000)
    lane_dir = tmp_path / "Data" / "Intensities" / "BaseCalls" / "L003"
    tiles = [1101, 1102, 1178]
    filters = {t: synth.filter_bytes(spec, 3, t) for t in tiles}
    lane_dir.mkdir(parents=True)
    for t in tiles:
        (lane_dir / ("s_3_%d.filter" % t)).write_bytes(synth.filter_file_bytes(filters[t]))
    for cyc in range(4):
        cdir = lane_dir / ("C%d.1" % (cyc + 1))
        cdir.mkdir()
        planes = {t: synth.plane_bytes(spec, 3, t, cyc) for t in tiles}
        (cdir / "L003_1.cbcl").write_bytes(
            synth.cbcl_file_bytes(planes, filters if excluded else None))
    rd = bcl.BCLReader(str(tmp_path))
    for t in tiles:
        tile = rd.get_tile(3, t)
        for cyc in range(4):
            raw = synth.plane_bytes(spec, 3, t, cyc)
            got = tile.read_plane(cyc)
            want_code = np.where(raw == 0, 4, raw & 3)
            if excluded:                                  # failed wells read as 'N' (:312-314)
                want_code = np.where(filters[t] & 1, want_code, 4)
            assert (np.where(got == 0, 4, got & 3) == want_code).all()
        seqs = tile.get_seqs(range(0, n, 97), 0, 4)
        assert all(len(s) == 4 for s, _ in seqs.values())
    with pytest.raises(AssertionError):                   # tile not in the cbcl table (:295)
        (lane_dir / "s_3_1150.filter").write_bytes(synth.filter_file_bytes(filters[1101]))
        rd.get_tile(3, 1150).read_plane(0)
