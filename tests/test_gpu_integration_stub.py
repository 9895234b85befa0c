"""The ctypes stub of INTEGRATION.md, run as written there (its own CDLL handle, only the
prototypes it declares, planes handed over in host memory) on a real run directory: the
`lane_dupl` it builds must be the one the unmodified reference handed to output_writer
(tests/golden/mid.json)."""
import ctypes
import gzip
import os

import numpy as np
import pytest

from helpers import GOLD, load_fixture
from well_duplicates_amd import _lib, bcl, synth
from well_duplicates_amd.targets import load_targets

pytestmark = pytest.mark.gpu


def test_integration_stub_rebuilds_lane_dupl(tmp_path):
    fx = load_fixture("mid")
    spec = synth.spec_from_dict(fx["spec"])
    all_cycles = sorted({c for run in fx["runs"] for a, b in run["cycles"] for c in range(a, b)})
    synth.write_run_dir(spec, str(tmp_path), fx["lanes"], fx["tiles"], all_cycles)

    # --- near the imports ---------------------------------------------------------------
    _wd = ctypes.CDLL(_lib.LIB_PATH)
    _wd.wd_create.restype = ctypes.c_void_p
    _wd.wd_create.argtypes = [ctypes.c_int]
    _wd.wd_destroy.argtypes = [ctypes.c_void_p]
    _wd.wd_strerror.restype = ctypes.c_char_p
    _wd.wd_set_targets.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 3
    _wd.wd_count_tiles.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                   ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    _EXC = {-1: ValueError, -2: IndexError, -3: AssertionError}

    def _ck(rc):
        if rc:
            raise _EXC.get(rc, RuntimeError)(_wd.wd_strerror(rc).decode())

    # --- once per run, after load_targets() ------------------------------------------------
    level = fx["levels"]
    targets = load_targets(os.path.join(GOLD, fx["targets_file"]), levels=level + 1, limit=fx["n_targets"])
    ctx = _wd.wd_create(0)
    assert ctx
    try:
        centre = np.array([t.get_centre() for t in targets], dtype=np.int32)
        rings = [[t.get_indices(l + 1) for l in range(level)] for t in targets]
        lens = np.array([[len(r) for r in tr] for tr in rings], dtype=np.int64)
        lvl_off = np.zeros((len(rings), level + 1), dtype=np.int64)
        lvl_off[:, 1:] = np.cumsum(lens, axis=1)
        lvl_off += np.concatenate([[0], np.cumsum(lens.sum(axis=1))[:-1]])[:, None]
        lvl_off = lvl_off.astype(np.int32)
        nbr = np.array([w for tr in rings for r in tr for w in r], dtype=np.int32)
        _ck(_wd.wd_set_targets(ctx, len(centre), level, centre.ctypes.data, lvl_off.ctypes.data, nbr.ctypes.data))

        bcl_reader = bcl.BCLReader(str(tmp_path))
        for run in fx["runs"]:
            if run.get("exception"):
                continue
            mode = {"eq": 0, "hamming": 1, "levenshtein": 2}[run["mode"]]
            k = run["k"]
            for lane_rec in run["lanes"]:
                lane = lane_rec["lane"]
                for tile, want in lane_rec["lane_dupl"].items():
                    # --- per tile -------------------------------------------------------
                    tile_bcl = bcl_reader.get_tile(lane, tile)
                    filt = np.fromfile(tile_bcl.filter_file, dtype=np.uint8, offset=12)
                    planes = []
                    for start, end in run["cycles"]:
                        for cyc in range(start, end):
                            with gzip.open(os.path.join(tile_bcl.data_dir, "C%i.1" % (cyc + 1),
                                                        tile_bcl.bcl_filename)) as fh:
                                raw = np.frombuffer(fh.read(), dtype=np.uint8)
                            assert raw[:4].view("<u4")[0] == tile_bcl.num_clusters
                            planes.append(np.ascontiguousarray(raw[4:]))
                    ptrs = (ctypes.c_void_p * len(planes))(*[p.ctypes.data for p in planes])
                    fptr = (ctypes.c_void_p * 1)(filt.ctypes.data)
                    block = np.zeros(1 + 5 * level, dtype=np.int64)
                    per_target = np.zeros((len(centre), level), dtype=np.uint32)
                    _ck(_wd.wd_count_tiles(ctx, 1, len(planes), mode, k, ptrs, fptr, tile_bcl.num_clusters,
                                           block.ctypes.data, per_target.ctypes.data))
                    got = [[[int(per_target[t, l]), int(lens[t, l])] for l in range(level)]
                           for t in range(len(centre)) if per_target[t, 0] != 0xFFFFFFFF]
                    assert got == want, (run["flags"], lane, tile)
    finally:
        _wd.wd_destroy(ctx)


def test_integration_stub_batch_ingest(tmp_path):
    """The second stub of INTEGRATION.md ("the gunzip loop on the GPU"), as written: planes and filters
    of a batch of tiles through wd_load_tile_files_batch into device memory, wd_count_tiles on the device
    pointers - the reference's lane_dupl again."""
    fx = load_fixture("mid")
    spec = synth.spec_from_dict(fx["spec"])
    all_cycles = sorted({c for run in fx["runs"] for a, b in run["cycles"] for c in range(a, b)})
    synth.write_run_dir(spec, str(tmp_path), fx["lanes"], fx["tiles"], all_cycles)
    _wd = ctypes.CDLL(_lib.LIB_PATH)
    _wd.wd_create.restype = ctypes.c_void_p
    _wd.wd_create.argtypes = [ctypes.c_int]
    _wd.wd_destroy.argtypes = [ctypes.c_void_p]
    _wd.wd_strerror.restype = ctypes.c_char_p
    _wd.wd_malloc.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
    _wd.wd_free.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    _wd.wd_set_targets.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 3
    _wd.wd_count_tiles.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                   ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    _wd.wd_load_tile_files_batch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p),
                                             ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint8), ctypes.c_int64,
                                             ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]

    def _ck(rc):
        if rc:
            raise RuntimeError(_wd.wd_strerror(rc).decode())

    level = fx["levels"]
    targets = load_targets(os.path.join(GOLD, fx["targets_file"]), levels=level + 1, limit=fx["n_targets"])
    ctx = _wd.wd_create(0)
    assert ctx
    try:
        centre, lvl_off, nbr = targets.to_csr(level)
        lens = np.diff(lvl_off.astype(np.int64), axis=1)
        _ck(_wd.wd_set_targets(ctx, len(centre), level, centre.ctypes.data, lvl_off.ctypes.data, nbr.ctypes.data))
        bcl_reader = bcl.BCLReader(str(tmp_path))
        run = next(r for r in fx["runs"] if not r.get("exception"))
        mode, k, cycles = {"eq": 0, "hamming": 1, "levenshtein": 2}[run["mode"]], run["k"], run["cycles"]
        for lane_rec in run["lanes"]:
            batch = [bcl_reader.get_tile(lane_rec["lane"], t) for t in lane_rec["lane_dupl"]]
            # --- as in INTEGRATION.md ---------------------------------------------------------
            n, L = batch[0].num_clusters, sum(e - s for s, e in cycles)
            stride = (n + 255) // 256 * 256
            dev = ctypes.c_void_p()
            _ck(_wd.wd_malloc(ctx, stride * (L + 1) * len(batch), ctypes.byref(dev)))
            paths, dsts, kinds = [], [], []
            for i, t in enumerate(batch):
                base = dev.value + i * stride * (L + 1)
                for j, cyc in enumerate(c for s, e in cycles for c in range(s, e)):
                    paths.append(os.path.join(t.data_dir, "C%i.1" % (cyc + 1), t.bcl_filename).encode())
                    dsts.append(base + j * stride)
                    kinds.append(0)
                paths.append(t.filter_file.encode())
                dsts.append(base + L * stride)
                kinds.append(1)
            rc = (ctypes.c_int * len(paths))()
            _ck(_wd.wd_load_tile_files_batch(ctx, len(paths), (ctypes.c_char_p * len(paths))(*paths),
                                             (ctypes.c_void_p * len(paths))(*dsts), (ctypes.c_uint8 * len(paths))(*kinds),
                                             n, 1, 16, rc))
            # --- the scan on the device pointers --------------------------------------------
            ptrs = (ctypes.c_void_p * (L * len(batch)))(*[dev.value + i * stride * (L + 1) + j * stride
                                                          for i in range(len(batch)) for j in range(L)])
            fptr = (ctypes.c_void_p * len(batch))(*[dev.value + i * stride * (L + 1) + L * stride for i in range(len(batch))])
            block = np.zeros((len(batch), 1 + 5 * level), dtype=np.int64)
            per_target = np.zeros((len(batch), len(centre), level), dtype=np.uint32)
            _ck(_wd.wd_count_tiles(ctx, len(batch), L, mode, k, ptrs, fptr, n, block.ctypes.data, per_target.ctypes.data))
            for i, (tile, want) in enumerate(lane_rec["lane_dupl"].items()):
                got = [[[int(per_target[i, t, l]), int(lens[t, l])] for l in range(level)]
                       for t in range(len(centre)) if per_target[i, t, 0] != 0xFFFFFFFF]
                assert got == want, (run["flags"], tile)
            _ck(_wd.wd_free(ctx, dev))
    finally:
        _wd.wd_destroy(ctx)
