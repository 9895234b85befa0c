"""Deterministic synthetic flowcell data (BCL planes, .filter bytes, s.locs geometry).

Every byte is a pure function of (seed, lane, tile, cycle, cluster) through a
counter-based 64-bit hash, so the same tile can be produced
  * point-wise by numpy for a handful of wells (CPU tests, the oracle),
  * as whole planes by numpy (golden-fixture generation: real .bcl.gz files),
  * as whole planes on the device by the `wd_synth_*` HIP kernels (bench, GPU tests)
without shipping any data.  `tests/test_synth.py` pins the numpy and device
generators against each other and against committed known-answer bytes.

Byte semantics follow the reference reader: 0 = no-call ('N'), otherwise
base = byte & 3 and quality = byte >> 2 (bcl_direct_reader.py:352-358);
filter bit 0 = pass (bcl_direct_reader.py:246).

The spec: 0.5 % no-calls, 70 % filter pass, 2 % of wells are
"planted" near-copies of the well 1, ROW or 2*ROW records earlier.  A planted well
copies the *raw* (un-planted) bytes of its source so there are no chains.  Variants:
exact copy, copy with one or two substitutions, or a copy shifted by one cycle
(an indel-like pair: Hamming distance large, Levenshtein distance 2).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass

import numpy as np

M64 = (1 << 64) - 1

# Multipliers (odd 64-bit constants).  The HIP generator in csrc/synth_kernels.inc uses the same.
K_SEED = 0x9E3779B97F4A7C15
K_LANE = 0xD1B54A32D192ED03
K_TILE = 0x8CB92BA72F3D8DD7
K_CYCLE = 0xDB4F0B9175AE2165
K_CLUSTER = 0xA24BAED4963EE407
SALT_PLANT = 0x5851F42D4C957F2D
SALT_FILTER = 0x2545F4914F6CDD1D

# Planted-copy variants, chosen by (g >> 24) & 7
VAR_EXACT_MAX = 4      # 0..4 exact copy
VAR_SUB1 = 5           # one substituted cycle
VAR_SUB2 = 6           # two substituted cycles
VAR_SHIFT = 7          # source read one cycle later
SUB_CYCLE_MOD = 128    # substituted cycles are drawn from [0, 128)


def mix64_int(x: int) -> int:
    """splitmix64 finaliser on a Python int."""
    x &= M64
    x ^= x >> 30
    x = (x * 0xBF58476D1CE4E5B9) & M64
    x ^= x >> 27
    x = (x * 0x94D049BB133111EB) & M64
    x ^= x >> 31
    return x


def mix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wraps modulo 2**64)."""
    x = x.astype(np.uint64, copy=True)
    x ^= x >> np.uint64(30)
    x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27)
    x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    return x


@dataclass(frozen=True)
class SynthSpec:
    """Parameters of one synthetic flowcell."""
    seed: int = 1
    n_clusters: int = 4309253          # 2743 rows x 1571 wells (HiSeq 4000 geometry)
    row: int = 1571                    # wells per row (offset of the well "one row up")
    nocall_per_64k: int = 328          # 0.5 %
    pass_per_64k: int = 45875          # 70 %
    plant_per_64k: int = 1311          # 2 %
    filter_noise: bool = False         # set filter bit 1 at random (tests `flag & 1`)
    dead_tiles: tuple = ()             # tile numbers whose every cluster fails the filter
    plant_far: bool = False            # copies from up to 5 rows / 3 wells away (outer levels)
    qual_levels: int = 39              # distinct quality values; 7-8 mimics binned qualities
                                       # (and makes the planes compress ~2x, like real .bcl.gz)

    def tile_key(self, lane: int, tile: int, salt: int) -> int:
        return mix64_int(self.seed * K_SEED + lane * K_LANE + tile * K_TILE + salt)

    def plane_key(self, lane: int, tile: int, cycle: int) -> int:
        return mix64_int(self.seed * K_SEED + lane * K_LANE + tile * K_TILE
                         + (cycle + 1) * K_CYCLE)


def _raw_bytes(spec: SynthSpec, plane_key: int, clusters: np.ndarray) -> np.ndarray:
    """Un-planted base-call byte of each cluster in one plane."""
    with np.errstate(over="ignore"):
        h = mix64(np.uint64(plane_key) + clusters.astype(np.uint64) * np.uint64(K_CLUSTER))
    nocall = (h & np.uint64(0xFFFF)) < np.uint64(spec.nocall_per_64k)
    base = (h >> np.uint64(16)) & np.uint64(3)
    qual = np.uint64(2) + ((h >> np.uint64(18)) & np.uint64(0xFFFF)) % np.uint64(spec.qual_levels)
    b = ((qual << np.uint64(2)) | base).astype(np.uint8)
    b[nocall] = 0
    return b


def plant_info(spec: SynthSpec, lane: int, tile: int, clusters: np.ndarray):
    """(source cluster, variant, sub cycle 1, sub cycle 2) of each cluster.

    source == cluster for wells that are not planted copies.
    """
    clusters = np.asarray(clusters, dtype=np.int64)
    with np.errstate(over="ignore"):
        g = mix64(np.uint64(spec.tile_key(lane, tile, SALT_PLANT))
                  + clusters.astype(np.uint64) * np.uint64(K_CLUSTER))
    planted = (g & np.uint64(0xFFFF)) < np.uint64(spec.plant_per_64k)
    if spec.plant_far:
        # 8 source offsets reaching honeycomb levels 1..5
        sel = ((g >> np.uint64(16)) & np.uint64(7)).astype(np.int64)
        table = np.array([1, 2, 3, spec.row, 2 * spec.row, 3 * spec.row, 4 * spec.row,
                          5 * spec.row], dtype=np.int64)
        delta = table[sel]
    else:
        sel = ((g >> np.uint64(16)) & np.uint64(0xFF)) % np.uint64(3)
        delta = np.where(sel == 0, 1, np.where(sel == 1, spec.row, 2 * spec.row)).astype(np.int64)
    src = clusters - delta
    planted &= src >= 0
    src = np.where(planted, src, clusters)
    variant = ((g >> np.uint64(24)) & np.uint64(7)).astype(np.int64)
    variant = np.where(planted, variant, 0)
    sub1 = ((g >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64) % SUB_CYCLE_MOD
    sub2 = ((g >> np.uint64(48)) & np.uint64(0xFFFF)).astype(np.int64) % SUB_CYCLE_MOD
    return src, variant, sub1, sub2


_PLANT_CACHE: dict = {}


def _plant_info_cached(spec: SynthSpec, lane: int, tile: int, clusters):
    """plant_info for a whole tile, kept for the next plane of the same tile."""
    if clusters is not None:
        return plant_info(spec, lane, tile, clusters)
    key = (spec, lane, tile)
    if key not in _PLANT_CACHE:
        _PLANT_CACHE.clear()
        src, variant, sub1, sub2 = plant_info(spec, lane, tile,
                                              np.arange(spec.n_clusters, dtype=np.int64))
        where = np.flatnonzero(variant | (src != np.arange(spec.n_clusters)))
        _PLANT_CACHE[key] = (where, src[where], variant[where], sub1[where], sub2[where])
    return _PLANT_CACHE[key]


def plane_bytes(spec: SynthSpec, lane: int, tile: int, cycle: int,
                clusters: np.ndarray | None = None) -> np.ndarray:
    """Base-call bytes of `clusters` (default: the whole plane) at 0-based `cycle`.

    `cycle` is the reference's 0-based base position: it lives in directory C<cycle+1>.1
    (bcl_direct_reader.py:201).
    """
    key_here = spec.plane_key(lane, tile, cycle)
    if clusters is None:
        # whole plane: hash every cluster once, then patch the ~2 % planted wells
        out = _raw_bytes(spec, key_here, np.arange(spec.n_clusters, dtype=np.int64))
        where, src, variant, sub1, sub2 = _plant_info_cached(spec, lane, tile, None)
        own = out[where]
    else:
        clusters = np.asarray(clusters, dtype=np.int64)
        src, variant, sub1, sub2 = plant_info(spec, lane, tile, clusters)
        own = _raw_bytes(spec, key_here, clusters)
        out = own.copy()
        where = np.arange(len(clusters))
    patched = _raw_bytes(spec, key_here, src)
    shift = variant == VAR_SHIFT
    if shift.any():
        patched[shift] = _raw_bytes(spec, spec.plane_key(lane, tile, cycle + 1), src[shift])
    subst = ((variant == VAR_SUB1) | (variant == VAR_SUB2)) & (sub1 == cycle)
    subst |= (variant == VAR_SUB2) & (sub2 == cycle)
    patched[subst] = own[subst]
    out[where] = patched
    return out


def filter_bytes(spec: SynthSpec, lane: int, tile: int,
                 clusters: np.ndarray | None = None) -> np.ndarray:
    """.filter payload bytes (bit 0 = pass)."""
    if clusters is None:
        clusters = np.arange(spec.n_clusters, dtype=np.int64)
    clusters = np.asarray(clusters, dtype=np.int64)
    with np.errstate(over="ignore"):
        f = mix64(np.uint64(spec.tile_key(lane, tile, SALT_FILTER))
                  + clusters.astype(np.uint64) * np.uint64(K_CLUSTER))
    b = ((f & np.uint64(0xFFFF)) < np.uint64(spec.pass_per_64k)).astype(np.uint8)
    if int(tile) in tuple(int(t) for t in spec.dead_tiles):
        b[:] = 0
    if spec.filter_noise:
        b |= (((f >> np.uint64(16)) & np.uint64(1)) << np.uint64(1)).astype(np.uint8)
    return b


# ---------------------------------------------------------------------------------------
# Geometry: a honeycomb s.locs (plan.md:30-36 gives the pitch)
# ---------------------------------------------------------------------------------------

def honeycomb_pixels(rows: int, cols: int):
    """Integer pixel coordinates (as the reference decodes them) of a rows x cols honeycomb.

    Cluster index = row * cols + col.  x pitch 20.5 px, odd rows offset by 10.25 px,
    row pitch 17.75 px, origin at pixel (1000, 1000).
    """
    r = np.repeat(np.arange(rows, dtype=np.int64), cols)
    c = np.tile(np.arange(cols, dtype=np.int64), rows)
    x = np.floor(1000.0 + c * 20.5 + (r & 1) * 10.25 + 0.5).astype(np.int64)
    y = np.floor(1000.0 + r * 17.75 + 0.5).astype(np.int64)
    return x, y


def slocs_bytes(x_pix: np.ndarray, y_pix: np.ndarray) -> bytes:
    """Serialise pixel coordinates as an s.locs file.

    Layout: int32, float32, uint32 n, then n x (float32 x, float32 y)
    (prepare_cluster_indexes.py:135-137); the reader decodes int(v * 10 + 1000.5)
    (prepare_cluster_indexes.py:110-112), so v = (pix - 1000) / 10.
    """
    n = len(x_pix)
    body = np.empty((n, 2), dtype="<f4")
    body[:, 0] = (x_pix - 1000) / 10.0
    body[:, 1] = (y_pix - 1000) / 10.0
    # Guard: the float32 round trip must decode to the same integer pixel.
    dec_x = (body[:, 0].astype(np.float64) * 10.0 + 1000.5).astype(np.int64)
    dec_y = (body[:, 1].astype(np.float64) * 10.0 + 1000.5).astype(np.int64)
    assert (dec_x == x_pix).all() and (dec_y == y_pix).all()
    return struct.pack("<ifI", 1, 1.0, n) + body.tobytes()


def filter_file_bytes(payload: np.ndarray) -> bytes:
    """.filter file: uint32 0, uint32 3, uint32 n, then n bytes (bcl_direct_reader.py:148-152)."""
    return struct.pack("<III", 0, 3, len(payload)) + payload.tobytes()


def bcl_file_bytes(payload: np.ndarray) -> bytes:
    """Uncompressed .bcl content: uint32 n + n bytes (bcl_direct_reader.py:333-338)."""
    return struct.pack("<I", len(payload)) + payload.tobytes()


def write_run_dir(spec: SynthSpec, run_dir: str, lanes, tiles, cycles, slocs: bytes | None = None,
                  compresslevel: int = 1) -> None:
    """Materialise a run directory the reference (and this package's CLI) can read.

    Layout (bcl_direct_reader.py:100, :124-129, :201-204; Snakefile.count_dups:168):
      <run>/Data/Intensities/s.locs
      <run>/Data/Intensities/BaseCalls/L00<lane>/s_<lane>_<tile>.filter
      <run>/Data/Intensities/BaseCalls/L00<lane>/C<cycle+1>.1/s_<lane>_<tile>.bcl.gz
    cycles: iterable of 0-based cycle numbers to write.
    """
    import gzip
    import os

    inten = os.path.join(run_dir, "Data", "Intensities")
    os.makedirs(inten, exist_ok=True)
    if slocs is not None:
        with open(os.path.join(inten, "s.locs"), "wb") as fh:
            fh.write(slocs)
    for lane in lanes:
        ldir = os.path.join(inten, "BaseCalls", "L%03d" % int(lane))
        os.makedirs(ldir, exist_ok=True)
        for tile in tiles:
            stem = "s_%d_%s" % (int(lane), tile)
            with open(os.path.join(ldir, stem + ".filter"), "wb") as fh:
                fh.write(filter_file_bytes(filter_bytes(spec, int(lane), int(tile))))
            for cyc in cycles:
                cdir = os.path.join(ldir, "C%d.1" % (cyc + 1))
                os.makedirs(cdir, exist_ok=True)
                payload = plane_bytes(spec, int(lane), int(tile), cyc)
                with gzip.open(os.path.join(cdir, stem + ".bcl.gz"), "wb",
                               compresslevel=compresslevel) as fh:
                    fh.write(bcl_file_bytes(payload))


def write_run_dir_cbcl(spec: SynthSpec, run_dir: str, lanes, tiles, cycles, excluded: bool = True,
                       slocs: bytes | None = None) -> None:
    """NovaSeq-style run directory: per cycle and lane one `L00<lane>_<surface>.cbcl` holding
    every tile of that surface (bcl_direct_reader.py:137, :255-325), plus the .filter files."""
    import os

    inten = os.path.join(run_dir, "Data", "Intensities")
    os.makedirs(inten, exist_ok=True)
    if slocs is not None:
        with open(os.path.join(inten, "s.locs"), "wb") as fh:
            fh.write(slocs)
    for lane in lanes:
        lname = "L%03d" % int(lane)
        ldir = os.path.join(inten, "BaseCalls", lname)
        os.makedirs(ldir, exist_ok=True)
        filters = {int(t): filter_bytes(spec, int(lane), int(t)) for t in tiles}
        for t in tiles:
            with open(os.path.join(ldir, "s_%d_%s.filter" % (int(lane), t)), "wb") as fh:
                fh.write(filter_file_bytes(filters[int(t)]))
        for cyc in cycles:
            cdir = os.path.join(ldir, "C%d.1" % (cyc + 1))
            os.makedirs(cdir, exist_ok=True)
            for surface in sorted({str(t)[0] for t in tiles}):
                mine = [int(t) for t in tiles if str(t)[0] == surface]
                planes = {t: plane_bytes(spec, int(lane), t, cyc) for t in mine}
                flt = {t: filters[t] for t in mine} if excluded else None
                with open(os.path.join(cdir, "%s_%s.cbcl" % (lname, surface)), "wb") as fh:
                    fh.write(cbcl_file_bytes(planes, flt))


def spec_to_dict(spec: SynthSpec) -> dict:
    d = dict(spec.__dict__)
    d["dead_tiles"] = list(spec.dead_tiles)
    return d


def spec_from_dict(d: dict) -> SynthSpec:
    d = dict(d)
    d["dead_tiles"] = tuple(d.get("dead_tiles", ()))
    return SynthSpec(**d)


def cbcl_file_bytes(tile_planes: dict, filters: dict | None = None, header_size: int = 5681,
                    compresslevel: int = 1) -> bytes:
    """One cycle's .cbcl file for several tiles (layout: bcl_direct_reader.py:255-325,
    cbcl_read.py:19-99): header '<HIBBI' (version 1, header size, 2 base bits, 2 quality
    bits, 4 bins), 4 x (uint32, uint32) bin table, uint32 tile count, per tile
    (tile, clusters, uncompressed size, compressed size), 1 byte excluded flag, padding up to
    header_size, then one gzip member per tile holding 2 wells per byte, low nibble first.

    tile_planes: {tile number: N plane bytes}; only (byte & 3) | (4 if byte else 0)... the
    nibble written is `byte & 0xF` with zero kept as no-call, so `nibble & 3` is the base.
    filters: if given, {tile: N filter bytes} and only passing wells are stored
    (excluded flag = 1, the NovaSeq default).
    """
    import gzip
    import struct as _s

    tiles = sorted(tile_planes)
    blocks, table = [], b""
    for t in tiles:
        plane = np.asarray(tile_planes[t], dtype=np.uint8)
        # a called base must stay non-zero in 4 bits: keep the base, set a quality bit
        nib = np.where(plane == 0, 0, (plane & 3) | 4).astype(np.uint8)
        if filters is not None:
            nib = nib[(np.asarray(filters[t]) & 1) == 1]
        n = nib.shape[0]
        if n % 2:
            nib = np.concatenate([nib, np.zeros(1, np.uint8)])
        packed = (nib[0::2] | (nib[1::2] << 4)).astype(np.uint8).tobytes()
        comp = gzip.compress(packed, compresslevel=compresslevel)
        blocks.append(comp)
        table += _s.pack("<IIII", int(t), n, len(packed), len(comp))
    head = _s.pack("<HIBBI", 1, header_size, 2, 2, 4)
    head += b"".join(_s.pack("<II", i, i) for i in range(4))
    head += _s.pack("<I", len(tiles)) + table + bytes([1 if filters is not None else 0])
    assert len(head) <= header_size
    head += b"\0" * (header_size - len(head))
    return head + b"".join(blocks)
