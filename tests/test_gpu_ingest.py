"""Native ingest (wd_load_bcl_gz / wd_load_filter / wd_gather_wells): files -> device memory."""
import gzip
import os
import struct
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from well_duplicates_amd import synth
from well_duplicates_amd.scanner import Scanner, TileBatch

pytestmark = pytest.mark.gpu

SPEC = synth.SynthSpec(seed=6, n_clusters=300007, row=517, nocall_per_64k=3000, plant_per_64k=9000)


@pytest.fixture(scope="module")
def sc():
    s = Scanner(0)
    yield s
    s.close()


@pytest.fixture(scope="module")
def run_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("ingest")
    synth.write_run_dir(SPEC, str(d), [1], ["1101", "1102"], list(range(20)), compresslevel=6)
    return str(d)


def _paths(run_dir, tile, cyc):
    base = os.path.join(run_dir, "Data", "Intensities", "BaseCalls", "L001")
    return (os.path.join(base, "C%d.1" % (cyc + 1), "s_1_%s.bcl.gz" % tile),
            os.path.join(base, "s_1_%s.filter" % tile))


def test_threaded_load_equals_spec(sc, run_dir):
    n = SPEC.n_clusters
    tb = TileBatch(sc, 2, 20, n)
    jobs = [(i, t, c) for i, t in enumerate(("1101", "1102")) for c in range(20)]
    with ThreadPoolExecutor(max_workers=8) as pool:
        list(pool.map(lambda j: sc.load_bcl_gz(_paths(run_dir, j[1], j[2])[0], tb.plane_ptr(j[0], j[2]), n), jobs))
    for i, t in enumerate(("1101", "1102")):
        sc.load_filter(_paths(run_dir, t, 0)[1], tb.filter_ptr(i), n)
        assert (tb.download_filter(i) == synth.filter_bytes(SPEC, 1, int(t))).all()
        for c in (0, 7, 19):
            assert (tb.download_plane(i, c) == synth.plane_bytes(SPEC, 1, int(t), c)).all()
    # gather: bytes of a few wells over all cycles
    wells = np.array([0, 5, 299999, 300006, 12345], np.int32)
    got = sc.gather_wells([tb.plane_ptr(1, c) for c in range(20)], wells, n)
    want = np.stack([synth.plane_bytes(SPEC, 1, 1102, c, wells) for c in range(20)], axis=1)
    assert (got == want).all()
    with pytest.raises(IndexError):
        sc.gather_wells([tb.plane_ptr(1, 0)], np.array([n], np.int32), n)
    tb.free()


def test_errors_and_odd_files(sc, run_dir, tmp_path):
    n = SPEC.n_clusters
    dst = sc.malloc(n + 64)
    with pytest.raises(FileNotFoundError):                       # bcl_direct_reader.py:207-216
        sc.load_bcl_gz(str(tmp_path / "nope.bcl.gz"), dst, n)
    with pytest.raises(AssertionError):                          # header != clusters (:338)
        sc.load_bcl_gz(_paths(run_dir, "1101", 0)[0], dst, n - 1)
    with pytest.raises(AssertionError):
        sc.load_filter(_paths(run_dir, "1101", 0)[1], dst, n + 1)
    # concatenated gzip members are one stream to gzip.open - and to the loader
    payload = synth.plane_bytes(SPEC, 1, 1101, 3)
    raw = synth.bcl_file_bytes(payload)
    p = tmp_path / "multi.bcl.gz"
    p.write_bytes(gzip.compress(raw[:1000]) + gzip.compress(raw[1000:200000]) + gzip.compress(raw[200000:]))
    sc.load_bcl_gz(str(p), dst, n)
    assert (sc.d2h(dst, n) == payload).all()
    # truncated payload: the reference dies with IndexError at slurped_file[idx]
    q = tmp_path / "short.bcl.gz"
    q.write_bytes(gzip.compress(raw[:-10]))
    with pytest.raises(IndexError):
        sc.load_bcl_gz(str(q), dst, n)
    # what gzip.open(..).read() raises in the reference (bcl_direct_reader.py:208-209), with the file named:
    # not gzip at all -> BadGzipFile
    r = tmp_path / "junk.bcl.gz"
    r.write_bytes(b"hello world" * 10)
    with pytest.raises(gzip.BadGzipFile, match="junk.bcl.gz"):
        sc.load_bcl_gz(str(r), dst, n)
    # the compressed stream ends early -> EOFError
    whole = gzip.compress(raw)
    t = tmp_path / "cut.bcl.gz"
    t.write_bytes(whole[:len(whole) // 2])
    with pytest.raises(EOFError, match="cut.bcl.gz"):
        sc.load_bcl_gz(str(t), dst, n)
    # bad data inside the stream -> zlib.error
    import zlib
    dmg = bytearray(whole)
    for pos in range(len(dmg) // 2, len(dmg) // 2 + 64):
        dmg[pos] ^= 0xFF
    u = tmp_path / "damaged.bcl.gz"
    u.write_bytes(bytes(dmg))
    with pytest.raises((zlib.error, gzip.BadGzipFile), match="damaged.bcl.gz"):
        sc.load_bcl_gz(str(u), dst, n)
    assert not issubclass(zlib.error, FileNotFoundError)     # (the CLI only falls back to .cbcl when the file is absent)
    # bad filter header version
    f = tmp_path / "s_1_9.filter"
    f.write_bytes(struct.pack("<III", 0, 2, 5) + b"\1" * 5)
    with pytest.raises(AssertionError):                          # :151
        sc.load_filter(str(f), dst, 5)
    sc.free(dst)


@pytest.mark.parametrize("excluded", [False, True])
def test_cbcl_native_equals_host_reader(sc, tmp_path, excluded):
    """wd_load_cbcl_tile (gunzip on the host, nibble expansion + excluded-wells ranks on the GPU)
    gives the same base codes and no-calls as the host reader of bcl_direct_reader.py:255-325."""
    from well_duplicates_amd import bcl
    n = 70001                                               # odd; 69 chunks of 1024 wells
    spec = synth.SynthSpec(seed=12, n_clusters=n, row=211, nocall_per_64k=5000, pass_per_64k=40000)
    tiles = ["1101", "1150", "2103"]
    cycles = list(range(3))
    synth.write_run_dir_cbcl(spec, str(tmp_path), [2], tiles, cycles, excluded=excluded)
    rd = bcl.BCLReader(str(tmp_path))
    tb = TileBatch(sc, len(tiles), len(cycles), n)
    handles = [rd.get_tile(2, t) for t in tiles]
    for i, h in enumerate(handles):
        sc.load_filter(h.filter_file, tb.filter_ptr(i), n)
    jobs = [(i, c) for i in range(len(tiles)) for c in cycles]
    with ThreadPoolExecutor(max_workers=6) as pool:
        list(pool.map(lambda j: sc.load_cbcl_tile(handles[j[0]].cbcl_path(j[1]), int(tiles[j[0]]),
                                                  tb.filter_ptr(j[0]), n, tb.plane_ptr(j[0], j[1])), jobs))
    for i, h in enumerate(handles):
        for c in cycles:
            got = tb.download_plane(i, c)
            want = h.read_plane(c)
            assert (np.where(got == 0, 4, got & 3) == np.where(want == 0, 4, want & 3)).all()
            if excluded:
                assert (got[(synth.filter_bytes(spec, 2, int(tiles[i])) & 1) == 0] == 0).all()
    with pytest.raises(AssertionError):                     # tile not in the table (:295)
        sc.load_cbcl_tile(handles[0].cbcl_path(0), 1199, tb.filter_ptr(0), n, tb.plane_ptr(0, 0))
    with pytest.raises(FileNotFoundError):
        sc.load_cbcl_tile(str(tmp_path / "nope.cbcl"), 1101, tb.filter_ptr(0), n, tb.plane_ptr(0, 0))
    bad = tmp_path / "bad.cbcl"
    raw = bytearray(open(handles[0].cbcl_path(0), "rb").read())
    raw[0] = 2                                              # version != 1 (:266)
    bad.write_bytes(bytes(raw))
    with pytest.raises(AssertionError):
        sc.load_cbcl_tile(str(bad), 1101, tb.filter_ptr(0), n, tb.plane_ptr(0, 0))
    tb.free()
