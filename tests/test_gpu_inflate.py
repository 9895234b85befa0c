"""GPU DEFLATE decoder (csrc/gpu_inflate.inc, wd_load_bcl_gz_batch) against Python's gzip module -
the reference's own decompressor (bcl_direct_reader.py:208-209: gzip.open(...).read())."""
import gzip
import os
import zlib

import numpy as np
import pytest

from helpers import REPO
from well_duplicates_amd import synth
from well_duplicates_amd.scanner import Scanner

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sc():
    s = Scanner(0)
    yield s
    s.close()


def _bcl(payload: np.ndarray, level: int = 6, **kw) -> bytes:
    return gzip.compress(synth.bcl_file_bytes(payload), compresslevel=level, **kw)


def _planes(n, kinds):
    """Payloads that stress different parts of the decoder."""
    rng = np.random.default_rng(42)
    out = {}
    for kind in kinds:
        if kind.startswith("q"):                        # base calls with that many quality values
            spec = synth.SynthSpec(seed=9, n_clusters=n, row=1000, qual_levels=int(kind[1:]), nocall_per_64k=2000)
            out[kind] = synth.plane_bytes(spec, 1, 1101, 7)
        elif kind == "noise":                           # incompressible: stored blocks
            out[kind] = rng.integers(0, 256, n, dtype=np.uint8)
        elif kind == "runs":                            # long matches, short distances, overlapping copies
            v = rng.integers(0, 4, n // 50 + 1, dtype=np.uint8).repeat(50)[:n] * 7 + 8
            out[kind] = v.astype(np.uint8)
        elif kind == "zeros":                           # a failed cycle: every well a no-call (the decoder declines, zlib decodes)
            out[kind] = np.zeros(n, np.uint8)
        elif kind == "period":                          # far matches with long lengths
            base = rng.integers(1, 255, 9000, dtype=np.uint8)
            out[kind] = np.tile(base, n // 9000 + 1)[:n]
        elif kind == "skew":                            # very uneven symbol frequencies: long and short codes
            p = 0.5 ** np.arange(1, 25)
            out[kind] = rng.choice(24, size=n, p=p / p.sum()).astype(np.uint8) + 1
        else:
            raise KeyError(kind)
    return out


def _load_and_check(sc, tmp_path, files, n, threads=4):
    """files: {name: (bytes on disk, expected payload or exception class)}"""
    names = list(files)
    paths = []
    for name in names:
        p = tmp_path / (name + ".bcl.gz")
        p.write_bytes(files[name][0])
        paths.append(str(p))
    stride = (n + 255) // 256 * 256
    buf = sc.malloc(stride * len(names) + 256)
    sc.memset(buf, 0xEE, stride * len(names) + 256)
    dsts = [buf + i * stride for i in range(len(names))]
    g0, h0 = sc.get_option("inflate_files_gpu"), sc.get_option("inflate_files_host")
    good = [i for i, nm in enumerate(names) if isinstance(files[nm][1], np.ndarray)]
    sc.load_bcl_gz_batch([paths[i] for i in good], [dsts[i] for i in good], n, threads=threads)
    for i in good:
        got = sc.d2h(dsts[i], n)
        assert (got == files[names[i]][1]).all(), names[i]
        if n < stride:                                  # nothing written past the plane
            assert (sc.d2h(dsts[i] + n, stride - n) == 0xEE).all(), names[i]
    for i, nm in enumerate(names):
        if i in good:
            continue
        with pytest.raises(files[nm][1], match=nm):
            sc.load_bcl_gz_batch([paths[i]], [dsts[i]], n, threads=1)
    by_gpu = sc.get_option("inflate_files_gpu") - g0
    by_host = sc.get_option("inflate_files_host") - h0
    sc.free(buf)
    return by_gpu, by_host


def test_small_window_form(sc, tmp_path):
    """A launch whose files all expand less than 1.75-fold is decoded by the four-wave kernel with the
    small window buffers (three files per CU): planes that compress like binned qualities, noise, and
    files with stretches that expand far more than the window holds (cut windows, or the host)."""
    sc.set_option("inflate_waves", 4)
    n = 400003
    pl = _planes(n, ["q39", "noise", "runs"])
    rng = np.random.default_rng(11)
    files = {"q39_l6": (_bcl(pl["q39"], 6), pl["q39"]), "noise_l6": (_bcl(pl["noise"], 6), pl["noise"])}
    for k in range(6):
        mixed = pl["noise"].copy()
        at = int(rng.integers(0, n - 70000))
        mixed[at:at + 40000] = pl["runs"][:40000]
        at = int(rng.integers(0, n - 70000))
        mixed[at:at + int(rng.integers(100, 30000))] = int(rng.integers(0, 256))
        files["mixed%d" % k] = (_bcl(mixed, 6 if k % 2 else 9), mixed)
    for name, (data, payload) in files.items():
        assert (len(payload) + 4) * 4 <= len(data) * 7, name      # what the loader's choice goes by
    try:
        by_gpu, by_host = _load_and_check(sc, tmp_path, files, n)
    finally:
        sc.set_option("inflate_waves", 0)
    assert by_gpu + by_host == len(files) and by_gpu >= 2, (by_gpu, by_host)


@pytest.mark.parametrize("waves", [1, 4, 8])
def test_batch_matches_gzip_module(sc, tmp_path, waves):
    """Every kernel variant (waves per file) on every kind of stream."""
    sc.set_option("inflate_waves", waves)
    n = 400003
    pl = _planes(n, ["q7", "q39", "q2", "noise", "runs", "period", "skew", "zeros"])
    files = {}
    for kind, payload in pl.items():
        for level in (1, 6, 9):
            files["%s_l%d" % (kind, level)] = (_bcl(payload, level), payload)
    # fixed Huffman codes only (Z_FIXED), and a stream with sync-flush points (empty stored blocks)
    co = zlib.compressobj(6, zlib.DEFLATED, 31, 8, zlib.Z_FIXED)
    raw = synth.bcl_file_bytes(pl["q7"])
    files["fixed"] = (co.compress(raw) + co.flush(), pl["q7"])
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    parts = [co.compress(raw[i:i + 50000]) + co.flush(zlib.Z_SYNC_FLUSH) for i in range(0, len(raw), 50000)]
    files["flushes"] = (b"".join(parts) + co.flush(), pl["q7"])
    # little expansion over the whole file (the launch takes the decoder's small-window form when all
    # its files are like that - see test_small_window_form) but stretches that expand a lot
    mixed = pl["noise"].copy()
    mixed[100000:160000] = pl["runs"][:60000]
    mixed[300000:330000] = 7
    files["mixed"] = (_bcl(mixed, 6), mixed)
    # a file name in the gzip header (FNAME), as `gzip file` writes it
    files["named"] = (gzip.compress(raw, 6)[:3] + b"\x08" + gzip.compress(raw, 6)[4:10] + b"s_1_1101.bcl\0"
                      + gzip.compress(raw, 6)[10:], pl["q7"])
    try:
        by_gpu, by_host = _load_and_check(sc, tmp_path, files, n)
    finally:
        sc.set_option("inflate_waves", 0)
    # the decoder took everything but (some of) the all-zero planes, where one piece of the stream
    # expands more than a window holds
    assert by_gpu + by_host == len(files) and by_host <= 3, (by_gpu, by_host)


def test_batch_decoded_on_the_gpu_at_full_size(sc, tmp_path):
    """A full-size plane (4.3 M wells), more files than one staging chunk holds."""
    n = 4309253
    sc.set_option("inflate_chunk_mb", 8)
    try:
        files = {}
        for c, q in enumerate((7, 7, 39, 7, 2, 7, 7)):
            spec = synth.SynthSpec(seed=21, n_clusters=n, row=1571, qual_levels=q)
            payload = synth.plane_bytes(spec, 2, 1205, c)
            files["c%d" % c] = (_bcl(payload, 6 if c % 2 else 1), payload)
        # two values in turn: every match (258 bytes at distance 2) copies from the one before it - 16 700
        # deep, past the decoder's turn budget: the host takes the file, the rest stay on the GPU
        ab = np.tile(np.array([7, 9], np.uint8), n // 2 + 1)[:n]
        files["chain"] = (_bcl(ab, 6), ab)
        by_gpu, by_host = _load_and_check(sc, tmp_path, files, n, threads=3)
        assert (by_gpu, by_host) == (len(files) - 1, 1)
    finally:
        sc.set_option("inflate_chunk_mb", 16)


def test_batch_reports_what_the_reference_raises(sc, tmp_path):
    n = 100001
    spec = synth.SynthSpec(seed=5, n_clusters=n, row=333, qual_levels=7)
    payload = synth.plane_bytes(spec, 1, 1101, 0)
    raw = synth.bcl_file_bytes(payload)
    whole = gzip.compress(raw, 6)
    dmg = bytearray(whole)
    for pos in range(len(dmg) // 2, len(dmg) // 2 + 64):
        dmg[pos] ^= 0xFF
    crc = bytearray(whole)
    crc[-6] ^= 1                                        # CRC-32 in the trailer
    flip = bytearray(whole)
    flip[len(flip) // 3] ^= 4                           # one bit: often still a valid stream, caught by the CRC
    files = {
        "fine": (whole, payload),
        "multi": (gzip.compress(raw[:1000]) + gzip.compress(raw[1000:60000]) + gzip.compress(raw[60000:]), payload),
        "junk": (b"hello world" * 10, gzip.BadGzipFile),
        "cut": (whole[:len(whole) // 2], EOFError),
        "notrailer": (whole[:-8], EOFError),
        "damaged": (bytes(dmg), (zlib.error, gzip.BadGzipFile)),
        "badcrc": (bytes(crc), (zlib.error, gzip.BadGzipFile)),
        "bitflip": (bytes(flip), (zlib.error, gzip.BadGzipFile, AssertionError, IndexError)),
        "short": (gzip.compress(raw[:-10]), IndexError),                          # fails at slurped_file[idx] in the reference
        "count": (gzip.compress(synth.bcl_file_bytes(payload[:-1]) + b"\0"), AssertionError),   # header != clusters (:338)
        "long": (gzip.compress(raw + b"\1" * 100), payload),                        # extra bytes after the plane are never indexed
        # ... but the reference reads them all, so a bad CRC behind an overlong stream is still an error
        "longbadcrc": ((lambda z: z[:-6] + bytes([z[-6] ^ 1]) + z[-5:])(gzip.compress(raw + b"\1" * 200000)),
                       (zlib.error, gzip.BadGzipFile)),
    }
    # every one of these must agree with what Python's gzip says about the same bytes
    for name, (data, want) in files.items():
        if isinstance(want, np.ndarray):
            assert gzip.decompress(data)[4:4 + n] == want.tobytes(), name
        elif want in (EOFError, gzip.BadGzipFile) or name in ("damaged", "badcrc", "longbadcrc"):
            with pytest.raises((EOFError, gzip.BadGzipFile, zlib.error)):
                gzip.decompress(data)
    _load_and_check(sc, tmp_path, files, n)
    # a missing file among good ones: reported per file
    p = tmp_path / "fine.bcl.gz"
    buf = sc.malloc(2 * n + 512)
    missing = sc.load_bcl_gz_batch([str(p), str(tmp_path / "nope.bcl.gz")], [buf, buf + (n + 255) // 256 * 256], n,
                                   threads=2, missing_ok=True)
    assert missing == [1]
    with pytest.raises(FileNotFoundError, match="nope"):
        sc.load_bcl_gz_batch([str(p), str(tmp_path / "nope.bcl.gz")], [buf, buf + (n + 255) // 256 * 256], n)
    sc.load_bcl_gz_batch([], [], n)
    sc.free(buf)


def test_corrupted_streams_end_with_a_verdict(sc, tmp_path):
    """Random damage: the kernel must come back (no hang, no write outside the plane) and the loader's
    answer must be the host decoder's: the same plane or an exception."""
    n = 60001
    spec = synth.SynthSpec(seed=8, n_clusters=n, row=250, qual_levels=7)
    payload = synth.plane_bytes(spec, 1, 1101, 1)
    whole = gzip.compress(synth.bcl_file_bytes(payload), 6)
    rng = np.random.default_rng(7)
    stride = (n + 255) // 256 * 256
    cases = 96
    buf = sc.malloc(stride * cases + 256)
    sc.memset(buf, 0xEE, stride * cases + 256)
    paths, datas = [], []
    for i in range(cases):
        m = bytearray(whole)
        kind = i % 4
        if kind == 0:
            for _ in range(int(rng.integers(1, 4))):
                m[int(rng.integers(10, len(m)))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            m = m[:int(rng.integers(1, len(m)))]
        elif kind == 2:
            at = int(rng.integers(10, len(m)))
            m[at:at + 32] = bytes(rng.integers(0, 256, min(32, len(m) - at), dtype=np.uint8))
        else:
            m[10 + int(rng.integers(0, 40))] ^= 1 << int(rng.integers(0, 8))
        p = tmp_path / ("case%03d.bcl.gz" % i)
        p.write_bytes(bytes(m))
        paths.append(str(p))
        datas.append(bytes(m))
    rcs_ok = 0
    for i in range(cases):                              # one at a time: each case gets its own verdict
        try:
            sc.load_bcl_gz_batch([paths[i]], [buf + i * stride], n, threads=1)
            ok = True
        except (EOFError, zlib.error, gzip.BadGzipFile, AssertionError, IndexError):
            ok = False
        try:
            ref = gzip.decompress(datas[i])
            ref_ok = len(ref) >= n + 4 and int.from_bytes(ref[:4], "little") == n
        except (EOFError, zlib.error, gzip.BadGzipFile):
            ref, ref_ok = None, False
        assert ok == ref_ok, (i, ok, ref_ok)
        if ok:
            rcs_ok += 1
            assert sc.d2h(buf + i * stride, n).tobytes() == ref[4:4 + n]
        if n < stride:
            assert (sc.d2h(buf + i * stride + n, stride - n) == 0xEE).all(), i
    sc.free(buf)


def test_filters_travel_with_the_batch(sc, tmp_path):
    """wd_load_tile_files_batch: .filter files in the same call (copied, not decoded), checked as
    wd_load_filter checks them (bcl_direct_reader.py:148-152, :236-240)."""
    import struct
    n = 50003
    spec = synth.SynthSpec(seed=4, n_clusters=n, row=211, qual_levels=7)
    stride = (n + 255) // 256 * 256
    buf = sc.malloc(6 * stride)
    sc.memset(buf, 0xEE, 6 * stride)
    gz, want = [], []
    for c in range(2):
        payload = synth.plane_bytes(spec, 1, 1101, c)
        p = tmp_path / ("c%d.bcl.gz" % c)
        p.write_bytes(_bcl(payload))
        gz.append(str(p))
        want.append(payload)
    filt = []
    for t in (1101, 1102):
        f = tmp_path / ("s_1_%d.filter" % t)
        f.write_bytes(synth.filter_file_bytes(synth.filter_bytes(spec, 1, t)))
        filt.append(str(f))
    sc.load_bcl_gz_batch(gz, [buf, buf + stride], n, threads=3,
                         filters=[(filt[0], buf + 2 * stride), (filt[1], buf + 3 * stride)])
    for i in range(2):
        assert (sc.d2h(buf + i * stride, n) == want[i]).all()
        assert (sc.d2h(buf + (2 + i) * stride, n) == synth.filter_bytes(spec, 1, 1101 + i)).all()
        assert (sc.d2h(buf + (2 + i) * stride + n, stride - n) == 0xEE).all()
    # only filters, no planes
    sc.load_bcl_gz_batch([], [], n, filters=[(filt[1], buf + 4 * stride)])
    assert (sc.d2h(buf + 4 * stride, n) == synth.filter_bytes(spec, 1, 1102)).all()
    bad = tmp_path / "s_1_9.filter"
    bad.write_bytes(struct.pack("<III", 0, 2, n) + b"\1" * n)            # version != 3 (:151)
    with pytest.raises(AssertionError, match="s_1_9"):
        sc.load_bcl_gz_batch(gz[:1], [buf], n, filters=[(str(bad), buf + 5 * stride)])
    with pytest.raises(AssertionError):                                  # cluster count (:236)
        sc.load_bcl_gz_batch([], [], n + 1, filters=[(filt[0], buf + 5 * stride)])
    with pytest.raises(FileNotFoundError, match="nope"):
        sc.load_bcl_gz_batch(gz[:1], [buf], n, missing_ok=True, filters=[(str(tmp_path / "nope.filter"), buf + 5 * stride)])
    sc.free(buf)


@pytest.mark.parametrize("excluded", [False, True])
def test_cbcl_blocks_through_the_gpu_decoder(sc, tmp_path, excluded):
    """wd_load_cbcl_batch = wd_load_cbcl_tile entry by entry (bcl_direct_reader.py:255-325): the tiles'
    gzip blocks inflated in one launch, expanded with the excluded-wells indirection."""
    from well_duplicates_amd import bcl
    from well_duplicates_amd.scanner import TileBatch
    n = 70001
    spec = synth.SynthSpec(seed=12, n_clusters=n, row=211, nocall_per_64k=5000, pass_per_64k=40000)
    tiles = ["1101", "1150", "2103"]
    cycles = list(range(4))
    synth.write_run_dir_cbcl(spec, str(tmp_path), [2], tiles, cycles, excluded=excluded)
    rd = bcl.BCLReader(str(tmp_path))
    handles = [rd.get_tile(2, t) for t in tiles]
    one = TileBatch(sc, len(tiles), len(cycles), n)
    many = TileBatch(sc, len(tiles), len(cycles), n)
    for tb in (one, many):
        sc.load_bcl_gz_batch([], [], n, filters=[(h.filter_file, tb.filter_ptr(i)) for i, h in enumerate(handles)])
    jobs = [(i, c) for i in range(len(tiles)) for c in cycles]
    for i, c in jobs:
        sc.load_cbcl_tile(handles[i].cbcl_path(c), int(tiles[i]), one.filter_ptr(i), n, one.plane_ptr(i, c))
    g0, h0 = sc.get_option("inflate_files_gpu"), sc.get_option("inflate_files_host")
    sc.load_cbcl_batch([(handles[i].cbcl_path(c), int(tiles[i]), many.filter_ptr(i), many.plane_ptr(i, c)) for i, c in jobs],
                       n, threads=3)
    assert sc.get_option("inflate_files_gpu") - g0 == len(jobs) and sc.get_option("inflate_files_host") == h0
    for i, c in jobs:
        assert (many.download_plane(i, c) == one.download_plane(i, c)).all(), (i, c)
    # errors are the single-entry loader's: a tile that is not in the table (:295), a missing file
    with pytest.raises(AssertionError):
        sc.load_cbcl_batch([(handles[0].cbcl_path(0), 1199, many.filter_ptr(0), many.plane_ptr(0, 0))], n)
    with pytest.raises(FileNotFoundError, match="nope"):
        sc.load_cbcl_batch([(handles[0].cbcl_path(0), 1101, many.filter_ptr(0), many.plane_ptr(0, 0)),
                            (str(tmp_path / "nope.cbcl"), 1101, many.filter_ptr(0), many.plane_ptr(0, 1))], n)
    # a damaged block goes to the host loader, which reports it
    raw = bytearray(open(handles[1].cbcl_path(1), "rb").read())
    for pos in range(len(raw) - 400, len(raw) - 300):
        raw[pos] ^= 0x5A
    bad = tmp_path / "damaged.cbcl"
    bad.write_bytes(bytes(raw))
    with pytest.raises((zlib.error, gzip.BadGzipFile, EOFError, IndexError, AssertionError)):
        sc.load_cbcl_batch([(str(bad), int(tiles[-1]), many.filter_ptr(2), many.plane_ptr(2, 1))], n)
    one.free()
    many.free()


def test_interleaved_layout_through_the_gpu_decoder(sc, tmp_path):
    """well_stride = 4: every plane decoded on the GPU lands in its byte lane of the group of four cycles
    (what wd_load_bcl_gz_strided does on the host)."""
    from well_duplicates_amd.scanner import TileBatch
    n, cycles = 50003, 6
    spec = synth.SynthSpec(seed=14, n_clusters=n, row=211, qual_levels=7)
    tb = TileBatch(sc, 2, cycles, n, interleave=4)
    paths, dsts, want = [], [], {}
    for i, t in enumerate((1101, 1102)):
        for c in range(cycles):
            payload = synth.plane_bytes(spec, 1, t, c)
            p = tmp_path / ("t%d_c%d.bcl.gz" % (t, c))
            p.write_bytes(_bcl(payload))
            paths.append(str(p))
            dsts.append(tb.plane_ptr(i, c))
            want[(i, c)] = payload
    g0 = sc.get_option("inflate_files_gpu")
    sc.load_bcl_gz_batch(paths, dsts, n, threads=3, well_stride=4)
    assert sc.get_option("inflate_files_gpu") - g0 == len(paths)
    for (i, c), payload in want.items():
        assert (tb.download_plane(i, c) == payload).all(), (i, c)
    # a file the decoder declines takes the host path into the same lane
    z = tmp_path / "zeros.bcl.gz"
    z.write_bytes(_bcl(np.zeros(n, np.uint8)))
    sc.load_bcl_gz_batch([str(z)], [tb.plane_ptr(1, 3)], n, well_stride=4)
    assert (tb.download_plane(1, 3) == 0).all() and (tb.download_plane(1, 2) == want[(1, 2)]).all()
    tb.free()


def test_batches_from_several_threads(sc, tmp_path):
    """Three callers at once on one context (the CLI keeps two batches in flight): the calls share the
    pinned ring and the streams, alternate between two arenas, and every plane must be the right one."""
    from concurrent.futures import ThreadPoolExecutor
    n, per = 120001, 9
    stride = (n + 255) // 256 * 256
    buf = sc.malloc(3 * per * stride + 256)
    sets = []
    for t in range(3):
        spec = synth.SynthSpec(seed=30 + t, n_clusters=n, row=400, qual_levels=7)
        paths, dsts, want = [], [], []
        for c in range(per):
            payload = synth.plane_bytes(spec, 1, 1101 + t, c)
            p = tmp_path / ("s%d_c%d.bcl.gz" % (t, c))
            p.write_bytes(_bcl(payload, 1 if c % 2 else 6))
            paths.append(str(p))
            dsts.append(buf + (t * per + c) * stride)
            want.append(payload)
        sets.append((paths, dsts, want))
    for rep in range(3):
        sc.memset(buf, 0, 3 * per * stride)
        with ThreadPoolExecutor(max_workers=3) as pool:
            list(pool.map(lambda s: sc.load_bcl_gz_batch(s[0], s[1], n, threads=2), sets))
        for paths, dsts, want in sets:
            for d, w in zip(dsts, want):
                assert (sc.d2h(d, n) == w).all()
    sc.free(buf)


def test_thread_creation_failure_is_an_error_code_not_an_abort(sc, tmp_path):
    """The batch entry points start reader threads inside extern "C": a thread that cannot be started
    must not end the caller's process (std::terminate).  `test_thread_limit` makes thread creation
    fail after n threads per crew: with n >= 1 the batch is loaded by the threads there are, with 0
    the call returns WD_ERR_NOMEM (MemoryError here) - and the context goes on working."""
    n = 100001
    spec = synth.SynthSpec(seed=5, n_clusters=n, row=333, qual_levels=7)
    payloads = [synth.plane_bytes(spec, 1, 1101, c) for c in range(6)]
    paths = []
    for c, pl in enumerate(payloads):
        p = tmp_path / ("c%d.bcl.gz" % c)
        p.write_bytes(gzip.compress(synth.bcl_file_bytes(pl), 6))
        paths.append(str(p))
    junk = tmp_path / "multi.bcl.gz"                      # two members: the host loader's turn (second crew)
    raw = synth.bcl_file_bytes(payloads[0])
    junk.write_bytes(gzip.compress(raw[:5000]) + gzip.compress(raw[5000:]))
    paths.append(str(junk))
    stride = (n + 255) // 256 * 256
    buf = sc.malloc(stride * len(paths) + 256)
    dsts = [buf + i * stride for i in range(len(paths))]
    try:
        for limit in (2, 1):
            sc.set_option("test_thread_limit", limit)
            sc.memset(buf, 0xEE, stride * len(paths))
            sc.load_bcl_gz_batch(paths, dsts, n, threads=256)
            for i, pl in enumerate(payloads + [payloads[0]]):
                assert (sc.d2h(dsts[i], n) == pl).all(), (limit, i)
        sc.set_option("test_thread_limit", 0)
        with pytest.raises(MemoryError):
            sc.load_bcl_gz_batch(paths, dsts, n, threads=256)
    finally:
        sc.set_option("test_thread_limit", -1)
    sc.load_bcl_gz_batch(paths, dsts, n, threads=256)      # 256 real threads, and the context still works
    assert (sc.d2h(dsts[3], n) == payloads[3]).all()
    sc.free(buf)


def test_real_thread_limit_in_a_child_process(tmp_path):
    """The same under a real RLIMIT_NPROC (an ordinary user's limit; root is exempt): the child loads a
    batch with threads=256 while the limit leaves room for a few - it must exit by itself with an
    answer (loaded, or an error code), never by abort."""
    import subprocess
    import sys
    if os.geteuid() == 0:
        pytest.skip("RLIMIT_NPROC does not bind root")
    code = r'''
import gzip, os, resource, sys
sys.path.insert(0, %r)
from well_duplicates_amd import synth
from well_duplicates_amd.scanner import Scanner
n = 50001
spec = synth.SynthSpec(seed=5, n_clusters=n, row=333, qual_levels=7)
paths = []
for c in range(8):
    p = os.path.join(%r, "c%%d.bcl.gz" %% c)
    open(p, "wb").write(gzip.compress(synth.bcl_file_bytes(synth.plane_bytes(spec, 1, 1101, c)), 6))
    paths.append(p)
with Scanner(0) as sc:
    stride = (n + 255) // 256 * 256
    buf = sc.malloc(stride * 8 + 256)
    dsts = [buf + i * stride for i in range(8)]
    sc.load_bcl_gz_batch(paths, dsts, n, threads=4)          # the runtime's own threads exist now
    uid = os.getuid()
    mine = 0
    for pid in os.listdir("/proc"):
        if pid.isdigit():
            try:
                st = open("/proc/%%s/status" %% pid).read()
            except OSError:
                continue
            if ("Uid:\t%%d" %% uid) in st:
                mine += int(st.split("Threads:")[1].split()[0])
    resource.setrlimit(resource.RLIMIT_NPROC, (mine + 3, resource.getrlimit(resource.RLIMIT_NPROC)[1]))
    try:
        sc.load_bcl_gz_batch(paths, dsts, n, threads=256)
        ok = all((sc.d2h(dsts[c], n) == synth.plane_bytes(spec, 1, 1101, c)).all() for c in range(8))
        print("LOADED" if ok else "WRONG")
    except MemoryError:
        print("ERROR CODE")
''' % (REPO, str(tmp_path))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, (res.returncode, res.stderr[-2000:])
    assert res.stdout.strip().splitlines()[-1] in ("LOADED", "ERROR CODE"), res.stdout


def test_forged_cbcl_tile_count_is_a_format_error(sc, tmp_path):
    """A .cbcl header whose tile count is absurd (4e9 tiles = a 64 GB table) is a format error
    (AssertionError, as the reference's header asserts), not an allocation (bcl_direct_reader.py:274-286)."""
    import struct
    n = 1000
    head = struct.pack("<HIBBI", 1, 5681, 2, 2, 4) + b"".join(struct.pack("<II", i, i) for i in range(4))
    forged = tmp_path / "forged.cbcl"
    forged.write_bytes(head + struct.pack("<I", 0xF0000000) + b"\0" * 6000)
    buf = sc.malloc(4096)
    for load in (lambda: sc.load_cbcl_batch([(str(forged), 1101, buf, buf + 2048)], n),
                 lambda: sc.load_cbcl_tile(str(forged), 1101, buf, n, buf + 2048)):
        with pytest.raises(AssertionError):
            load()
    sc.free(buf)


def test_a_tiles_filter_error_comes_before_its_planes(sc, tmp_path):
    """The reference opens a tile's .filter before its cycle files: when both are bad, the filter's
    error is the one raised (bcl_direct_reader.py:124-132, :236 before :208)."""
    n = 20001
    spec = synth.SynthSpec(seed=5, n_clusters=n, row=333)
    good = tmp_path / "good.bcl.gz"
    good.write_bytes(gzip.compress(synth.bcl_file_bytes(synth.plane_bytes(spec, 1, 1101, 0)), 6))
    junk = tmp_path / "junk.bcl.gz"
    junk.write_bytes(b"not gzip at all" * 20)
    okf = tmp_path / "ok.filter"
    okf.write_bytes(synth.filter_file_bytes(synth.filter_bytes(spec, 1, 1101)))
    badf = tmp_path / "bad.filter"
    badf.write_bytes(synth.filter_file_bytes(synth.filter_bytes(spec, 1, 1101))[:-7])
    stride = (n + 255) // 256 * 256
    buf = sc.malloc(stride * 6)
    d = [buf + i * stride for i in range(6)]
    # tile 0: junk plane, good filter; tile 1: good plane, bad filter -> tile 0's plane error first
    with pytest.raises(gzip.BadGzipFile):
        sc.load_bcl_gz_batch([str(junk), str(good)], d[:2], n, filters=[(str(okf), d[2]), (str(badf), d[3])],
                             tile_of=[0, 1, 0, 1])
    # tile 0: junk plane AND bad filter -> the filter's assertion
    with pytest.raises(AssertionError):
        sc.load_bcl_gz_batch([str(junk), str(good)], d[:2], n, filters=[(str(badf), d[2]), (str(okf), d[3])],
                             tile_of=[0, 1, 0, 1])
    # without the map: filters first
    with pytest.raises(AssertionError):
        sc.load_bcl_gz_batch([str(junk), str(good)], d[:2], n, filters=[(str(okf), d[2]), (str(badf), d[3])])
    sc.free(buf)


def test_failed_cycles_are_decoded_beside_the_batch_not_after_it(sc, tmp_path):
    """A fifth of the planes are failed cycles (every well a no-call: the stream expands a
    thousandfold, which the GPU decoder declines).  The reader thread that meets such a file decodes
    it at once on the host, while the rest of the batch is still being read, copied and inflated - so
    three batches in flight stay three batches in flight instead of each ending in a serial host tail.
    Every plane right, every failed cycle counted as decoded early, both layouts."""
    from concurrent.futures import ThreadPoolExecutor
    n, per = 200003, 20
    stride = (n + 255) // 256 * 256
    buf = sc.malloc(3 * per * stride + 256)
    sets, n_zero = [], 0
    for t in range(3):
        spec = synth.SynthSpec(seed=40 + t, n_clusters=n, row=400, qual_levels=7)
        paths, dsts, want = [], [], []
        for c in range(per):
            payload = synth.plane_bytes(spec, 1, 1101 + t, c)
            if c % 5 == 2:
                payload = np.zeros(n, np.uint8)
                n_zero += 1
            p = tmp_path / ("s%d_c%d.bcl.gz" % (t, c))
            p.write_bytes(_bcl(payload, 6))
            paths.append(str(p))
            dsts.append(buf + (t * per + c) * stride)
            want.append(payload)
        sets.append((paths, dsts, want))
    e0, h0, g0 = (sc.get_option(o) for o in ("inflate_files_early", "inflate_files_host", "inflate_files_gpu"))
    sc.memset(buf, 0xEE, 3 * per * stride)
    with ThreadPoolExecutor(max_workers=3) as pool:
        list(pool.map(lambda s: sc.load_bcl_gz_batch(s[0], s[1], n, threads=4), sets))
    for paths, dsts, want in sets:
        for d, w in zip(dsts, want):
            assert (sc.d2h(d, n) == w).all()
    assert sc.get_option("inflate_files_early") - e0 == n_zero == 12
    assert sc.get_option("inflate_files_host") - h0 == n_zero             # nothing was left for a tail
    assert sc.get_option("inflate_files_gpu") - g0 == 3 * per - n_zero
    sc.free(buf)
    # the interleaved layout: the host path scatters the plane into its byte lane itself
    group = sc.malloc(4 * stride + 256)
    sc.memset(group, 0xEE, 4 * stride)
    paths, dsts, want = sets[0][0][:4], [group + q for q in range(4)], sets[0][2][:4]      # cycles 0..3, cycle 2 failed
    sc.load_bcl_gz_batch(paths, dsts, n, threads=2, well_stride=4)
    got = sc.d2h(group, 4 * n).reshape(n, 4)
    for q in range(4):
        assert (got[:, q] == want[q]).all(), q
    sc.free(group)
