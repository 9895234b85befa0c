"""Run-directory reader feeding the device scan: whole planes, not per-read strings.

Host-side counterpart of the reference's bcl_direct_reader.py for this path.  The reference
turns every wanted cluster into a Python string, one base at a time
(bcl_direct_reader.py:347-361); the device wants the raw plane bytes, so this reader stops
after decompression: `Tile.read_plane(cycle)` returns the N payload bytes of one cycle and
`Tile.read_filter()` the N filter bytes.  Byte semantics are unchanged and are applied on
the device: 0 = no-call, else base = byte & 3 (:352-358); filter bit 0 = pass (:246).

Same directory conventions and the same exception classes as the reference:
  RuntimeError        no .filter file for the tile                       (:131-132)
  AssertionError      bad .filter header / plane length != filter length (:151, :236, :338)
  FileNotFoundError   neither <tile>.bcl.gz nor the lane's .cbcl exists  (:207-216)
CBCL (NovaSeq) planes are expanded to one byte per well (the 4-bit record value, whose low
two bits are the base and whose zero value is a no-call, :316-325), including the
excluded-wells indirection through the filter (:303-314).
"""
from __future__ import annotations

import gzip
import os
import re
import struct
import zlib

import numpy as np

SEQUENCE = 0     # bcl_direct_reader.py:54
QUAL_FLAG = 1    # bcl_direct_reader.py:55


def _gunzip(raw: bytes) -> bytes:
    """gzip -> bytes with one C call (zlib releases the GIL, so tiles decompress in
    parallel threads); multi-member files fall back to the gzip module."""
    d = zlib.decompressobj(16 + zlib.MAX_WBITS)
    data = d.decompress(raw)
    if d.unused_data:
        return gzip.decompress(raw)
    return data


class BCLReader:
    """One run directory (the one holding Data/ and RunInfo.xml), bcl_direct_reader.py:57-105."""

    def __init__(self, location="."):
        basecalls = os.listdir(os.path.join(location, "Data", "Intensities", "BaseCalls"))
        self.lanes = [d for d in basecalls if re.match(r"L\d\d\d$", d)]
        self.location = location
        self._listings = {}            # lane directory -> (names, number of cycle directories), listed once

    def get_tile(self, lane, tile):
        lane_dir = str(lane)
        if lane_dir not in self.lanes:
            lane_dir = "L%03d" % int(lane_dir)
        data_dir = os.path.join(self.location, "Data", "Intensities", "BaseCalls", lane_dir)
        # (the reference lists the directory for every tile, :124, :141; a lane has hundreds of tiles)
        def listed():
            names = os.listdir(data_dir)
            return names, len([f for f in names if re.match(r"C\d+.1$", f)])
        if data_dir not in self._listings:
            self._listings[data_dir] = listed()
        try:
            return Tile(data_dir, tile, self._listings[data_dir])
        except RuntimeError:                     # not in the listing: the directory may have grown since
            self._listings[data_dir] = listed()
            return Tile(data_dir, tile, self._listings[data_dir])


class Tile:
    """One tile's files (bcl_direct_reader.py:108-156)."""

    def __init__(self, data_dir, tile, listing=None):
        self.data_dir = data_dir
        self.tile = tile
        self.bcl_filename = None
        if listing is None:
            names = os.listdir(data_dir)
            listing = (names, len([f for f in names if re.match(r"C\d+.1$", f)]))
        listing, n_cycle_dirs = listing
        key = "_%s" % tile
        for name in listing:
            # the reference's pattern (:124-129), tried only on names that can match it
            m = re.match("(.+_%s).filter" % tile, name) if key in name else None
            if m:
                self.bcl_filename = m.group(1) + ".bcl.gz"
                self.filter_file = os.path.join(data_dir, name)
                break
        if not self.bcl_filename:
            raise RuntimeError("Cannot find a .filter file for tile %s" % tile)
        # "L00<lane>_<surface>.cbcl", surface = first digit of the tile id (:137)
        self.cbcl_filename = "%s_%s.cbcl" % (os.path.basename(data_dir), str(tile)[0])
        self.num_cycles = n_cycle_dirs
        with open(self.filter_file, "rb") as fh:
            head = struct.unpack("<III", fh.read(12))
        assert tuple(head[0:2]) == (0, 3)
        self.num_clusters = head[2]
        self._filter = None
        self._pass_index = None

    # ------------------------------------------------------------------ filter
    def read_filter(self) -> np.ndarray:
        """The N raw filter bytes (bit 0 = pass)."""
        if self._filter is None:
            with open(self.filter_file, "rb") as fh:
                head = fh.read(12)
                assert tuple(struct.unpack("<III", head)) == (0, 3, self.num_clusters)
                body = np.frombuffer(fh.read(), dtype=np.uint8)
            # struct.unpack('<NB') in the reference insists on exactly N bytes (:240)
            if body.shape[0] != self.num_clusters:
                raise struct.error("filter file holds %d bytes, header says %d"
                                   % (body.shape[0], self.num_clusters))
            self._filter = body
        return self._filter

    def _passing_wells(self) -> np.ndarray:
        if self._pass_index is None:
            self._pass_index = np.flatnonzero(self.read_filter() & 1)
        return self._pass_index

    # ------------------------------------------------------------------ planes
    def plane_path(self, cycle: int) -> str:
        """Path of the .bcl.gz of 0-based `cycle` (it may not exist: NovaSeq runs have .cbcl)."""
        return os.path.join(self.data_dir, "C%i.1" % (cycle + 1), self.bcl_filename)

    def cbcl_path(self, cycle: int) -> str:
        """Path of the lane/surface .cbcl of 0-based `cycle` (NovaSeq layout, :137)."""
        return os.path.join(self.data_dir, "C%i.1" % (cycle + 1), self.cbcl_filename)

    def read_plane(self, cycle: int) -> np.ndarray:
        """N base-call bytes of 0-based `cycle` (directory C<cycle+1>.1, :201)."""
        cycle_dir = os.path.join(self.data_dir, "C%i.1" % (cycle + 1))
        cycle_file = os.path.join(cycle_dir, self.bcl_filename)
        try:
            with open(cycle_file, "rb") as fh:
                raw = fh.read()
        except FileNotFoundError:
            with open(os.path.join(cycle_dir, self.cbcl_filename), "rb") as fh:
                return self._plane_from_cbcl(fh)
        data = _gunzip(raw)
        assert struct.unpack("<I", data[:4])[0] == self.num_clusters        # :338
        plane = np.frombuffer(data, dtype=np.uint8, offset=4)
        if plane.shape[0] < self.num_clusters:
            raise IndexError("index out of range")      # the reference fails at slurped_file[idx]
        return plane[:self.num_clusters]

    def _plane_from_cbcl(self, fh) -> np.ndarray:
        """One tile's block of a .cbcl file -> one byte per well (:255-325)."""
        h_version, h_size, h_basebits, h_qbits, h_bins = struct.unpack("<HIBBI", fh.read(12))
        assert h_version == 1
        assert h_size > 32
        assert h_basebits == 2
        assert h_qbits == 2
        assert h_bins == 4
        tail = fh.read(h_bins * 4 * 2 + 4)
        tile_count, = struct.unpack("<I", tail[-4:])
        table = fh.read(tile_count * 16 + 1)
        excluded = bool(table[-1])
        offset = h_size
        tile_as_int = int(self.tile)
        t_number = t_usize = None
        for t in range(tile_count):
            t_number, _t_clusters, t_usize, t_csize = struct.unpack("<IIII", table[t * 16:(t + 1) * 16])
            if t_number == tile_as_int:
                break
            offset += t_csize
        assert t_number == tile_as_int
        fh.seek(offset)
        block = gzip.GzipFile(fileobj=fh, mode="rb").read(t_usize)
        packed = np.frombuffer(block, dtype=np.uint8)
        nib = np.empty(packed.shape[0] * 2, dtype=np.uint8)
        nib[0::2] = packed & 0x0F          # even well index: low bits (:319-321)
        nib[1::2] = packed >> 4            # odd well index: high bits (:316-318)
        n = self.num_clusters
        if not excluded:
            if nib.shape[0] < n:
                raise IndexError("index out of range")
            return nib[:n]
        # only passing wells are stored; the others stay 'N' (:303-314)
        plane = np.zeros(n, dtype=np.uint8)
        passing = self._passing_wells()
        if nib.shape[0] < passing.shape[0]:
            raise IndexError("index out of range")
        plane[passing] = nib[:passing.shape[0]]
        return plane

    # ------------------------------------------------------------------ reference-style API
    def get_seqs(self, cluster_indices, start=0, end=None):
        """{idx: (sequence string, passed filter)} as the reference (:158-220); host only,
        for inspection and tests - the scan path never builds strings."""
        if end is None:
            end = self.num_cycles
        keys = sorted({int(i) for i in cluster_indices})
        if keys and keys[-1] >= self.num_clusters:
            raise IndexError("Requested cluster %i is out of range.  Highest on this tile is %i."
                             % (keys[-1], self.num_clusters - 1))
        if keys and keys[0] < 0:
            raise IndexError("Requested cluster %i is a negative number." % keys[0])
        idx = np.asarray(keys, dtype=np.int64)
        flags = (self.read_filter()[idx] & 1).astype(bool) if keys else []
        lut = np.frombuffer(b"NACGT", dtype="S1")
        cols = []
        for cycle in range(start, end):
            b = self.read_plane(cycle)[idx]
            cols.append(np.where(b == 0, 0, (b & 3) + 1))
        if cols:
            mat = lut[np.stack(cols, axis=1)]
            seqs = [row.tobytes().decode() for row in mat]
        else:
            seqs = [""] * len(keys)
        return {k: (s, bool(f)) for k, s, f in zip(keys, seqs, flags)}
