"""Python face of the device scan path (thin: everything heavy is in libwelldup.so).

`Scanner` wraps one `wd_ctx` (one GPU).  `TileBatch` keeps the BCL planes and filter bytes
of a list of tiles resident in HBM in the layout the kernels like best: one slab
[tile][cycle][N padded to 256 B], so consecutive cycle planes of a tile are a constant
stride apart and every plane starts 256-byte aligned.
"""
from __future__ import annotations

import ctypes
import os
from typing import Iterable, List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import MODE_EQ, MODE_HAMMING, MODE_LEVENSHTEIN, INVALID_TARGET  # noqa: F401

PLANE_ALIGN = 256


def compare_mode(edit_distance: int, hamming: bool):
    """Map the reference's -e / --hamming flags (count_well_duplicates.py:200, :258) to
    (mode, k).  -e 0 is string equality under either metric."""
    if edit_distance == 0:
        return MODE_EQ, 0
    return (MODE_HAMMING if hamming else MODE_LEVENSHTEIN), int(edit_distance)


def _raise(lib, ctx, rc: int, path: Optional[str] = None):
    """Raise what the reference would: the exception classes of include/welldup.h's table.
    `path` names the file a loader was reading (one of thousands per batch)."""
    detail = lib.wd_last_error(ctx).decode() if ctx else ""
    msg = "%s%s" % (lib.wd_strerror(rc).decode(), (": " + detail) if detail else "")
    if path is not None:
        msg = "%s: %s" % (msg, path)
    if rc == _lib.ERR_INDEX:
        raise IndexError(msg)            # bcl_direct_reader.py:186-192
    if rc == _lib.ERR_EMPTY_LEVEL:
        raise AssertionError(msg)        # count_well_duplicates.py:249
    if rc == _lib.ERR_ARG:
        raise ValueError(msg)
    if rc == _lib.ERR_NOMEM:
        raise MemoryError(msg)
    if rc == _lib.ERR_IO:
        raise FileNotFoundError(msg)      # bcl_direct_reader.py:207-216
    if rc == _lib.ERR_TRUNCATED:
        raise EOFError(msg)               # gzip.open(..).read() on a file that ends early (:208-209)
    if rc == _lib.ERR_CORRUPT:
        # ... on bad data: BadGzipFile when it is not a gzip file at all, else zlib.error
        import gzip
        import zlib
        magic = b""
        try:
            if path is not None and not path.endswith(".cbcl"):
                with open(path, "rb") as fh:
                    magic = fh.read(2)
        except OSError:
            pass
        if magic and magic != b"\x1f\x8b":
            raise gzip.BadGzipFile(msg)
        raise zlib.error(msg)
    if rc == _lib.ERR_FORMAT:
        raise AssertionError(msg)         # bcl_direct_reader.py:151, :236, :338
    raise RuntimeError(msg)


class Scanner:
    """One GPU context: resident targets + scan calls."""

    def __init__(self, device: int = -1):
        self._lib = _lib.load()
        self._ctx = self._lib.wd_create(device)
        if not self._ctx:
            rc = self._lib.wd_create_status()
            raise RuntimeError("wd_create(%d) failed: %s" % (device, self._lib.wd_strerror(rc).decode()))
        self.T = 0
        self.levels = 0
        self._owned = set()

    # ------------------------------------------------------------------ plumbing
    def _ck(self, rc: int):
        if rc != _lib.OK:
            _raise(self._lib, self._ctx, rc)

    def close(self, exiting: bool = False):
        """Destroys the context.  exiting=True: the process ends right after (the CLI's case) - the library
        then only waits for the device and leaves the freeing to the driver ("fast_exit", include/welldup.h)."""
        if self._ctx:
            if exiting:
                self._lib.wd_set_option(self._ctx, b"fast_exit", 1)
            else:
                for p in list(self._owned):
                    self._lib.wd_free(self._ctx, p)
            self._owned.clear()
            self._lib.wd_destroy(self._ctx)
            self._ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name: str, value: int):
        self._ck(self._lib.wd_set_option(self._ctx, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        v = ctypes.c_int64()
        self._ck(self._lib.wd_get_option(self._ctx, name.encode(), ctypes.byref(v)))
        return v.value

    def set_stream(self, hip_stream: Optional[int]):
        self._ck(self._lib.wd_set_stream(self._ctx, ctypes.c_void_p(hip_stream or 0)))

    def synchronize(self):
        self._ck(self._lib.wd_synchronize(self._ctx))

    def malloc(self, nbytes: int) -> int:
        p = ctypes.c_void_p()
        self._ck(self._lib.wd_malloc(self._ctx, int(nbytes), ctypes.byref(p)))
        self._owned.add(p.value)
        return p.value

    def free(self, ptr: int):
        if ptr in self._owned:
            self._owned.discard(ptr)
            self._ck(self._lib.wd_free(self._ctx, ctypes.c_void_p(ptr)))

    def h2d(self, dst: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        self._ck(self._lib.wd_memcpy_h2d(self._ctx, ctypes.c_void_p(dst),
                                         arr.ctypes.data_as(ctypes.c_void_p), arr.nbytes))

    def d2h(self, src: int, nbytes: int, dtype=np.uint8) -> np.ndarray:
        out = np.empty(int(nbytes) // np.dtype(dtype).itemsize, dtype=dtype)
        self._ck(self._lib.wd_memcpy_d2h(self._ctx, out.ctypes.data_as(ctypes.c_void_p),
                                         ctypes.c_void_p(src), out.nbytes))
        return out

    def memset(self, dst: int, value: int, nbytes: int):
        self._ck(self._lib.wd_memset(self._ctx, ctypes.c_void_p(dst), value, int(nbytes)))

    # ------------------------------------------------------------------ targets
    def set_targets(self, centre, lvl_off, nbr):
        """CSR as AllTargets.to_csr(): centre[T], lvl_off[T, levels+1], nbr[P] (int32)."""
        centre = np.ascontiguousarray(centre, dtype=np.int32)
        lvl_off = np.ascontiguousarray(lvl_off, dtype=np.int32)
        nbr = np.ascontiguousarray(nbr, dtype=np.int32)
        if lvl_off.ndim != 2 or lvl_off.shape[0] != centre.shape[0]:
            raise ValueError("lvl_off must be [T, levels+1]")
        T, levels = centre.shape[0], lvl_off.shape[1] - 1
        if T and int(lvl_off[:, -1].max()) > nbr.shape[0]:
            raise ValueError("lvl_off points past the end of nbr")
        self._ck(self._lib.wd_set_targets(
            self._ctx, T, levels, centre.ctypes.data_as(ctypes.c_void_p),
            lvl_off.ctypes.data_as(ctypes.c_void_p), nbr.ctypes.data_as(ctypes.c_void_p)))
        self.T, self.levels = T, levels

    def targets_from_coords(self, x, y, centres=None, levels=5, max_dists=None):
        """Neighbour rings on the device (prepare_cluster_indexes.py:38-78 semantics) for the
        given centre wells, or for every well when `centres` is None.  The result becomes the
        scanner's targets; returns (T, P)."""
        from .cluster_indexes import max_dists_for
        x = np.ascontiguousarray(x, dtype=np.int32)
        y = np.ascontiguousarray(y, dtype=np.int32)
        md = np.ascontiguousarray(max_dists if max_dists is not None else max_dists_for(levels),
                                  dtype=np.int32)
        if md.shape[0] != levels + 1:
            raise ValueError("max_dists must hold levels + 1 radii")
        c = None if centres is None else np.ascontiguousarray(centres, dtype=np.int32)
        P = ctypes.c_int64()
        self._ck(self._lib.wd_targets_from_coords(
            self._ctx, x.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p),
            x.shape[0], c.ctypes.data_as(ctypes.c_void_p) if c is not None else None,
            0 if c is None else c.shape[0], levels, md.ctypes.data_as(ctypes.c_void_p),
            ctypes.byref(P)))
        self.T = x.shape[0] if c is None else c.shape[0]
        self.levels = levels
        return self.T, P.value

    def get_targets(self):
        """Download the resident targets as (centre, lvl_off, nbr) int32 arrays."""
        T, lv, P = ctypes.c_int(), ctypes.c_int(), ctypes.c_int64()
        self._ck(self._lib.wd_targets_info(self._ctx, ctypes.byref(T), ctypes.byref(lv), ctypes.byref(P)))
        centre = np.zeros(T.value, dtype=np.int32)
        lvl_off = np.zeros((T.value, lv.value + 1), dtype=np.int32)
        nbr = np.zeros(P.value, dtype=np.int32)
        self._ck(self._lib.wd_get_targets(self._ctx, centre.ctypes.data_as(ctypes.c_void_p),
                                          lvl_off.ctypes.data_as(ctypes.c_void_p),
                                          nbr.ctypes.data_as(ctypes.c_void_p)))
        return centre, lvl_off, nbr

    # ------------------------------------------------------------------ scan
    @staticmethod
    def _tables(planes, filters, L):
        n_tiles = len(filters)
        flat = [int(p) for tile in planes for p in tile]
        if len(flat) != n_tiles * L:
            raise ValueError("planes must hold n_tiles x L pointers")
        pt = (ctypes.c_void_p * max(1, len(flat)))(*flat)
        ft = (ctypes.c_void_p * max(1, n_tiles))(*[int(f) for f in filters])
        return pt, ft

    def count_tiles(self, planes: Sequence[Sequence[int]], filters: Sequence[int], n_clusters: int,
                    mode: int, k: int, per_target: bool = False, tables=None, L=None):
        """Synchronous scan of n_tiles tiles.

        planes[i][c]: device address of tile i's c-th scanned cycle plane (N bytes);
        filters[i]: device address of its filter bytes.  Returns (blocks, per_target):
        blocks int64 [n_tiles, 1 + 5*levels] as documented in include/welldup.h;
        per_target uint32 [n_tiles, T, levels] or None.
        """
        n_tiles = len(filters)
        if L is None:
            L = len(planes[0]) if n_tiles else 0
        pt, ft = tables if tables is not None else self._tables(planes, filters, L)
        blocks = np.zeros((n_tiles, 1 + 5 * self.levels), dtype=np.int64)
        pto = np.zeros((n_tiles, self.T, self.levels), dtype=np.uint32) if per_target else None
        self._ck(self._lib.wd_count_tiles(
            self._ctx, n_tiles, L, mode, k, pt, ft, int(n_clusters),
            blocks.ctypes.data_as(ctypes.c_void_p),
            pto.ctypes.data_as(ctypes.c_void_p) if per_target else None))
        return blocks, pto

    def scan_async(self, tables, n_tiles: int, L: int, n_clusters: int, mode: int, k: int,
                   out_tile_dev: int, out_per_target_dev: Optional[int] = None):
        pt, ft = tables
        self._ck(self._lib.wd_scan_async(
            self._ctx, n_tiles, L, mode, k, pt, ft, int(n_clusters),
            ctypes.c_void_p(out_tile_dev), ctypes.c_void_p(out_per_target_dev or 0)))

    def scan_status(self):
        self._ck(self._lib.wd_scan_status(self._ctx))

    # ------------------------------------------------------------------ ingest
    def load_bcl_gz(self, path: str, dst: int, n_clusters: int, well_stride: int = 1):
        """gunzip a .bcl.gz straight into device memory (thread-safe, releases the GIL);
        well_stride = 4 writes the plane into its byte lane of an interleaved group."""
        rc = self._lib.wd_load_bcl_gz_strided(self._ctx, os.fsencode(path), ctypes.c_void_p(dst), int(n_clusters),
                                              int(well_stride))
        if rc != _lib.OK:
            _raise(self._lib, None, rc, path)

    def load_bcl_gz_batch(self, paths: Sequence[str], dsts: Sequence[int], n_clusters: int, threads: int = 16,
                          missing_ok: bool = False, filters: Sequence = (), well_stride: int = 1,
                          tile_of: Optional[Sequence[int]] = None):
        """Many .bcl.gz files -> device planes, inflated on the GPU (wd_load_bcl_gz_batch: host threads
        only read the compressed files; one wave per file decodes).  Raises what load_bcl_gz raises
        for the first file that fails; with missing_ok the files that do not exist are returned
        (as indices) instead, for the caller to look for a .cbcl.  filters: [(path, dst)] of the
        tiles' .filter files, loaded in the same call (a missing one always raises)."""
        n_gz = len(paths)
        paths = list(paths) + [f[0] for f in filters]
        dsts = list(dsts) + [f[1] for f in filters]
        n = len(paths)
        enc = [os.fsencode(p) for p in paths]
        c_paths = (ctypes.c_char_p * max(1, n))(*enc)
        c_dsts = (ctypes.c_void_p * max(1, n))(*[int(d) for d in dsts])
        kinds = (ctypes.c_uint8 * max(1, n))(*([0] * n_gz + [1] * (n - n_gz)))
        rcs = (ctypes.c_int * max(1, n))()
        rc = self._lib.wd_load_tile_files_batch(self._ctx, n, c_paths, c_dsts, kinds, int(n_clusters), int(well_stride),
                                                int(threads), rcs)
        if rc != _lib.OK and not any(rcs[i] != _lib.OK for i in range(n)):
            _raise(self._lib, None, rc)         # the call itself failed (no memory, no thread, HIP), not a file
        missing = []
        # which failure is reported: the reference meets a tile's .filter before that tile's cycle files
        # (bcl_direct_reader.py:124-132, :195 before :200-216) and the tiles one after the other; with
        # `tile_of` (tile of every plane, then of every filter) that order is kept, without it the
        # filters come first
        if tile_of is not None:
            order = sorted(range(n), key=lambda i: (tile_of[i], i < n_gz, i))
        else:
            order = list(range(n_gz, n)) + list(range(n_gz))
        for i in order:
            if rcs[i] == _lib.OK:
                continue
            if rcs[i] == _lib.ERR_IO and missing_ok and i < n_gz:
                missing.append(i)
                continue
            _raise(self._lib, None, rcs[i], paths[i])
        missing.sort()
        return missing

    def load_filter(self, path: str, dst: int, n_clusters: int):
        rc = self._lib.wd_load_filter(self._ctx, os.fsencode(path), ctypes.c_void_p(dst), int(n_clusters))
        if rc != _lib.OK:
            _raise(self._lib, None, rc, path)

    def load_cbcl_tile(self, path: str, tile: int, filter_dev: int, n_clusters: int, dst: int, well_stride: int = 1):
        """One tile's block of a NovaSeq .cbcl file -> byte plane on the device (thread-safe); well_stride = 4:
        into its byte lane of an interleaved group (dst = TileBatch.plane_ptr of an interleaved batch)."""
        rc = self._lib.wd_load_cbcl_tile_strided(self._ctx, os.fsencode(path), int(tile), ctypes.c_void_p(filter_dev),
                                                 int(n_clusters), ctypes.c_void_p(dst), int(well_stride))
        if rc != _lib.OK:
            _raise(self._lib, None, rc, path)

    def load_cbcl_batch(self, entries: Sequence, n_clusters: int, threads: int = 16, well_stride: int = 1):
        """entries: [(cbcl path, tile number, filter pointer, plane pointer)] - the tiles' blocks are
        inflated on the GPU in one launch and expanded (wd_load_cbcl_batch); the filters must be
        loaded.  Raises what load_cbcl_tile raises for the first entry that fails."""
        n = len(entries)
        c_paths = (ctypes.c_char_p * max(1, n))(*[os.fsencode(e[0]) for e in entries])
        c_tiles = (ctypes.c_int * max(1, n))(*[int(e[1]) for e in entries])
        c_filt = (ctypes.c_void_p * max(1, n))(*[int(e[2]) for e in entries])
        c_dst = (ctypes.c_void_p * max(1, n))(*[int(e[3]) for e in entries])
        rcs = (ctypes.c_int * max(1, n))()
        rc = self._lib.wd_load_cbcl_batch_strided(self._ctx, n, c_paths, c_tiles, c_filt, c_dst, int(n_clusters),
                                                  int(well_stride), int(threads), rcs)
        for i in range(n):
            if rcs[i] != _lib.OK:
                _raise(self._lib, None, rcs[i], entries[i][0])
        if rc != _lib.OK:
            _raise(self._lib, None, rc)         # the call itself failed, not an entry

    def gather_wells(self, plane_ptrs: Sequence[int], idx, n_clusters: int) -> np.ndarray:
        """uint8 [len(idx), L]: bytes of the given wells over the L planes."""
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        L = len(plane_ptrs)
        out = np.zeros((idx.shape[0], L), dtype=np.uint8)
        tbl = (ctypes.c_void_p * max(1, L))(*[int(p) for p in plane_ptrs])
        self._ck(self._lib.wd_gather_wells(self._ctx, tbl, L, idx.ctypes.data_as(ctypes.c_void_p),
                                           idx.shape[0], int(n_clusters),
                                           out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def gather_wells_batch(self, tb: "TileBatch", tile: int, idx) -> np.ndarray:
        """gather_wells over every cycle of one tile of a TileBatch, in the batch's layout."""
        self.set_option("well_stride", tb.interleave)
        try:
            return self.gather_wells([tb.plane_ptr(tile, c) for c in range(tb.L)], idx, tb.N)
        finally:
            self.set_option("well_stride", 1)

    # ------------------------------------------------------------------ dup log / profile
    def hitlog_enable(self, capacity: int):
        self._ck(self._lib.wd_hitlog_enable(self._ctx, int(capacity)))

    def hitlog_fetch(self, max_records: int):
        """-> (records of the last scan as a structured array, total number found).  The count is read
        first, so the host buffer is as large as the records there are, not as the log could hold."""
        total = ctypes.c_int64()
        self._ck(self._lib.wd_hitlog_fetch(self._ctx, None, 0, ctypes.byref(total)))
        # (the library copies at most what the device log held: records beyond its capacity were counted,
        # not kept - the caller sees that in total > len(records))
        n = max(0, min(total.value, int(max_records), self.get_option("hitlog_capacity")))
        dt = np.dtype([("tile", "<i4"), ("target", "<i4"), ("slot", "<i4"), ("dist", "<i4")])
        recs = np.zeros(n, dtype=dt)
        if n:
            self._ck(self._lib.wd_hitlog_fetch(self._ctx, recs.ctypes.data_as(ctypes.c_void_p), n, ctypes.byref(total)))
        return recs, total.value

    def profile_get(self):
        ms = ctypes.c_double()
        n = ctypes.c_int64()
        self._ck(self._lib.wd_profile_get(self._ctx, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def profile_reset(self):
        self._ck(self._lib.wd_profile_reset(self._ctx))

    def stream_read_gbs(self, dev_ptr: int, nbytes: int, passes: int = 5) -> float:
        """GB/s of a kernel that only reads `nbytes` of device memory (wd_stream_read_probe): what this box's
        HBM gives a stream - the yardstick beside a scan kernel's rate."""
        ms = ctypes.c_double()
        self._ck(self._lib.wd_stream_read_probe(self._ctx, ctypes.c_void_p(dev_ptr), nbytes, passes, ctypes.byref(ms)))
        return nbytes / (ms.value * 1e-3) / 1e9

    def last_kernel(self) -> str:
        """Template name of the compare kernel the last scan launched (wd_last_kernel)."""
        return self._lib.wd_last_kernel(self._ctx).decode()

    # ------------------------------------------------------------------ multi-GPU
    def comm_unique_id(self) -> bytes:
        buf = ctypes.create_string_buffer(_lib.UNIQUE_ID_BYTES)
        rc = self._lib.wd_comm_unique_id(buf)
        if rc != _lib.OK:
            _raise(self._lib, None, rc)
        return buf.raw

    def comm_init(self, rank: int, world: int, unique_id: bytes):
        assert len(unique_id) == _lib.UNIQUE_ID_BYTES
        self._ck(self._lib.wd_comm_init(self._ctx, rank, world, unique_id))

    def allreduce_counts(self, buf_dev: int, n: int):
        self._ck(self._lib.wd_allreduce_counts(self._ctx, ctypes.c_void_p(buf_dev), n))

    def comm_destroy(self):
        self._ck(self._lib.wd_comm_destroy(self._ctx))

    # ------------------------------------------------------------------ synthetic data
    def _spec_c(self, spec, tile: int):
        return _lib.SynthSpecC(spec.seed, spec.n_clusters, spec.row, spec.nocall_per_64k,
                               spec.pass_per_64k, spec.plant_per_64k, int(spec.filter_noise),
                               int(int(tile) in tuple(int(t) for t in spec.dead_tiles)),
                               int(spec.plant_far), int(spec.qual_levels))

    def synth_plane(self, dst: int, spec, lane: int, tile: int, cycle: int):
        s = self._spec_c(spec, tile)
        self._ck(self._lib.wd_synth_plane(self._ctx, ctypes.c_void_p(dst), ctypes.byref(s),
                                          int(lane), int(tile), int(cycle)))

    def synth_filter(self, dst: int, spec, lane: int, tile: int):
        s = self._spec_c(spec, tile)
        self._ck(self._lib.wd_synth_filter(self._ctx, ctypes.c_void_p(dst), ctypes.byref(s),
                                           int(lane), int(tile)))


class TileBatch:
    """BCL planes + filter bytes of a list of tiles, resident in HBM.

    Layout: planes[tile][cycle][N_pad] and filters[tile][N_pad], N_pad = N rounded up to
    256 bytes - or, with interleave=4, planes[tile][cycle // 4][N_pad][4]: the four cycles of a
    group side by side per well (include/welldup.h, wd_interleave4), which the equality /
    Hamming scan of sampled targets reads in half the cache lines.  Fill with `fill_synthetic`
    (device generator) or `upload_tile` (host bytes).
    """

    def __init__(self, scanner: Scanner, n_tiles: int, L: int, n_clusters: int, interleave: int = 1,
                 reuse: Optional["TileBatch"] = None):
        """reuse: a batch that is done with - its buffers are taken over when they are large enough
        (freeing device memory waits for every kernel in flight), else freed."""
        assert interleave in (1, 4)
        self.sc = scanner
        self.n_tiles, self.L, self.N = n_tiles, L, n_clusters
        self.interleave = interleave
        self.n_pad = (n_clusters + PLANE_ALIGN - 1) // PLANE_ALIGN * PLANE_ALIGN
        self.groups = (L + interleave - 1) // interleave          # plane slots per tile
        self.slot_bytes = self.n_pad * interleave
        self.plane_bytes = n_tiles * self.groups * self.slot_bytes
        self.filter_bytes = n_tiles * self.n_pad
        tmp_bytes = 4 * self.n_pad if interleave == 4 else 0
        if (reuse is not None and reuse.sc is scanner and reuse.d_planes and reuse._cap[0] >= self.plane_bytes
                and reuse._cap[1] >= self.filter_bytes and reuse._cap[2] >= tmp_bytes):
            self.d_planes, self.d_filters, self.d_tmp, self._cap = reuse.d_planes, reuse.d_filters, reuse.d_tmp, reuse._cap
            reuse.d_planes = reuse.d_filters = reuse.d_tmp = 0
        else:
            if reuse is not None:
                reuse.free()
            self.d_planes = scanner.malloc(max(1, self.plane_bytes))
            self.d_filters = scanner.malloc(max(1, self.filter_bytes))
            self.d_tmp = scanner.malloc(tmp_bytes) if tmp_bytes else 0
            self._cap = (self.plane_bytes, self.filter_bytes, tmp_bytes)
        self.tables = Scanner._tables(self.plane_ptrs(), self.filter_ptrs(), L)

    def plane_ptr(self, tile: int, cycle: int) -> int:
        """Address of well 0 of the cycle (wells are `interleave` bytes apart)."""
        g, sub = divmod(cycle, self.interleave)
        return self.d_planes + (tile * self.groups + g) * self.slot_bytes + sub

    def _put_plane(self, tile: int, cycle: int, produce):
        """produce(dst) writes one plain N-byte plane at dst; lands it in this batch's layout."""
        if self.interleave == 1:
            produce(self.plane_ptr(tile, cycle))
            return
        sub = cycle % 4
        produce(self.d_tmp + sub * self.n_pad)
        if sub == 3 or cycle == self.L - 1:                       # the group is complete
            src = (ctypes.c_void_p * 4)(*[self.d_tmp + i * self.n_pad if i <= sub else None for i in range(4)])
            self.sc._ck(self.sc._lib.wd_interleave4(self.sc._ctx, src, self.N,
                                                    ctypes.c_void_p(self.plane_ptr(tile, cycle - sub))))
            self.sc.synchronize()                                 # d_tmp is reused by the next group

    def filter_ptr(self, tile: int) -> int:
        return self.d_filters + tile * self.n_pad

    def plane_ptrs(self) -> List[List[int]]:
        return [[self.plane_ptr(i, c) for c in range(self.L)] for i in range(self.n_tiles)]

    def filter_ptrs(self) -> List[int]:
        return [self.filter_ptr(i) for i in range(self.n_tiles)]

    def fill_synthetic(self, spec, lane_tile: Sequence, cycles: Sequence[int]):
        """lane_tile: [(lane, tile number)] per batch slot; cycles: the L 0-based cycles."""
        assert len(lane_tile) == self.n_tiles and len(cycles) == self.L
        assert spec.n_clusters == self.N
        for i, (lane, tile) in enumerate(lane_tile):
            self.sc.synth_filter(self.filter_ptr(i), spec, lane, tile)
            for c, cyc in enumerate(cycles):
                self._put_plane(i, c, lambda dst, cyc=cyc: self.sc.synth_plane(dst, spec, lane, tile, cyc))
        self.sc.synchronize()

    def upload_tile(self, slot: int, planes: Iterable[np.ndarray], filt: np.ndarray):
        for c, p in enumerate(planes):
            assert p.shape[0] == self.N
            self._put_plane(slot, c, lambda dst, p=p: self.sc.h2d(dst, np.ascontiguousarray(p, dtype=np.uint8)))
        assert filt.shape[0] == self.N
        self.sc.h2d(self.filter_ptr(slot), np.ascontiguousarray(filt, dtype=np.uint8))

    def download_plane(self, slot: int, cycle: int) -> np.ndarray:
        if self.interleave == 1:
            return self.sc.d2h(self.plane_ptr(slot, cycle), self.N)
        g, sub = divmod(cycle, 4)
        group = self.sc.d2h(self.plane_ptr(slot, 4 * g), 4 * self.N)
        return np.ascontiguousarray(group[sub::4])

    def download_filter(self, slot: int) -> np.ndarray:
        return self.sc.d2h(self.filter_ptr(slot), self.N)

    def count(self, mode: int, k: int, per_target: bool = False):
        self.sc.set_option("well_stride", self.interleave)
        try:
            return self.sc.count_tiles(None, self.filter_ptrs(), self.N, mode, k, per_target,
                                       tables=self.tables, L=self.L)
        finally:
            self.sc.set_option("well_stride", 1)

    def free(self):
        for ptr in (self.d_planes, self.d_filters, self.d_tmp):
            if ptr:
                self.sc.free(ptr)
        self.d_planes = self.d_filters = self.d_tmp = 0
