"""ctypes binding of libwelldup.so (C ABI: include/welldup.h).

The library is built in-tree by `build()` (hipcc --offload-arch=gfx950) and must exist:
there is no CPU fallback for the scan path - `load()` raises if the shared object is
missing or a symbol the header declares cannot be resolved.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(REPO, "include")
# WELLDUP_LIB selects another build of the same ABI (kernel A/B experiments)
LIB_PATH = os.environ.get("WELLDUP_LIB") or os.path.join(HERE, "libwelldup.so")

OK = 0
ERR_ARG, ERR_INDEX, ERR_EMPTY_LEVEL, ERR_HIP, ERR_NOMEM, ERR_STATE, ERR_UNSUPPORTED, ERR_COMM, \
    ERR_NO_WELLS, ERR_IO, ERR_FORMAT, ERR_CORRUPT, ERR_TRUNCATED = -1, -2, -3, -4, -5, -6, -7, -8, -9, -10, -11, -12, -13
MODE_EQ, MODE_HAMMING, MODE_LEVENSHTEIN = 0, 1, 2
MAX_LEVELS = 32
INVALID_TARGET = 0xFFFFFFFF
UNIQUE_ID_BYTES = 128


class Hit(ctypes.Structure):
    _fields_ = [("tile", ctypes.c_int32), ("target", ctypes.c_int32),
                ("slot", ctypes.c_int32), ("dist", ctypes.c_int32)]


class SynthSpecC(ctypes.Structure):
    _fields_ = [("seed", ctypes.c_uint64), ("n_clusters", ctypes.c_int64), ("row", ctypes.c_int64),
                ("nocall_per_64k", ctypes.c_uint32), ("pass_per_64k", ctypes.c_uint32),
                ("plant_per_64k", ctypes.c_uint32), ("filter_noise", ctypes.c_uint32),
                ("tile_dead", ctypes.c_uint32), ("plant_far", ctypes.c_uint32),
                ("qual_levels", ctypes.c_uint32)]


_vp = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64
_sz = ctypes.c_size_t
_pp = ctypes.POINTER(ctypes.c_void_p)

# name -> (restype, argtypes); every symbol include/welldup.h declares
PROTOTYPES = {
    "wd_version": (_i, []),
    "wd_build_id": (ctypes.c_char_p, []),
    "wd_strerror": (ctypes.c_char_p, [_i]),
    "wd_last_error": (ctypes.c_char_p, [_vp]),
    "wd_create": (_vp, [_i]),
    "wd_create_status": (_i, []),
    "wd_destroy": (None, [_vp]),
    "wd_set_stream": (_i, [_vp, _vp]),
    "wd_synchronize": (_i, [_vp]),
    "wd_set_option": (_i, [_vp, ctypes.c_char_p, _i64]),
    "wd_get_option": (_i, [_vp, ctypes.c_char_p, ctypes.POINTER(_i64)]),
    "wd_malloc": (_i, [_vp, _sz, _pp]),
    "wd_free": (_i, [_vp, _vp]),
    "wd_memcpy_h2d": (_i, [_vp, _vp, _vp, _sz]),
    "wd_memcpy_d2h": (_i, [_vp, _vp, _vp, _sz]),
    "wd_memset": (_i, [_vp, _vp, _i, _sz]),
    "wd_set_targets": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "wd_targets_from_coords": (_i, [_vp, _vp, _vp, _i64, _vp, _i64, _i, _vp, ctypes.POINTER(_i64)]),
    "wd_targets_info": (_i, [_vp, ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_i64)]),
    "wd_get_targets": (_i, [_vp, _vp, _vp, _vp]),
    "wd_count_tiles": (_i, [_vp, _i, _i, _i, _i, _pp, _pp, _i64, _vp, _vp]),
    "wd_scan_async": (_i, [_vp, _i, _i, _i, _i, _pp, _pp, _i64, _vp, _vp]),
    "wd_scan_status": (_i, [_vp]),
    "wd_load_bcl_gz": (_i, [_vp, ctypes.c_char_p, _vp, _i64]),
    "wd_load_bcl_gz_strided": (_i, [_vp, ctypes.c_char_p, _vp, _i64, _i]),
    "wd_load_filter": (_i, [_vp, ctypes.c_char_p, _vp, _i64]),
    "wd_load_bcl_gz_batch": (_i, [_vp, _i, ctypes.POINTER(ctypes.c_char_p), _pp, _i64, _i, ctypes.POINTER(_i)]),
    "wd_load_tile_files_batch": (_i, [_vp, _i, ctypes.POINTER(ctypes.c_char_p), _pp, ctypes.POINTER(ctypes.c_uint8), _i64,
                                      _i, _i, ctypes.POINTER(_i)]),
    "wd_load_cbcl_tile": (_i, [_vp, ctypes.c_char_p, _i, _vp, _i64, _vp]),
    "wd_load_cbcl_batch": (_i, [_vp, _i, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(_i), _pp, _pp, _i64, _i,
                                ctypes.POINTER(_i)]),
    "wd_load_cbcl_tile_strided": (_i, [_vp, ctypes.c_char_p, _i, _vp, _i64, _vp, _i]),
    "wd_load_cbcl_batch_strided": (_i, [_vp, _i, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(_i), _pp, _pp, _i64, _i, _i,
                                        ctypes.POINTER(_i)]),
    "wd_interleave4": (_i, [_vp, ctypes.POINTER(ctypes.c_void_p), _i64, _vp]),
    "wd_gunzip": (_i, [_vp, ctypes.c_size_t, _vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), _i]),
    "wd_gather_wells": (_i, [_vp, _pp, _i, _vp, _i64, _i64, _vp]),
    "wd_hitlog_enable": (_i, [_vp, _i64]),
    "wd_hitlog_fetch": (_i, [_vp, _vp, _i64, ctypes.POINTER(_i64)]),
    "wd_profile_get": (_i, [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_i64)]),
    "wd_profile_reset": (_i, [_vp]),
    "wd_stream_read_probe": (_i, [_vp, _vp, ctypes.c_size_t, _i, ctypes.POINTER(ctypes.c_double)]),
    "wd_last_kernel": (ctypes.c_char_p, [_vp]),
    "wd_comm_unique_id": (_i, [_vp]),
    "wd_comm_init": (_i, [_vp, _i, _i, _vp]),
    "wd_allreduce_counts": (_i, [_vp, _vp, _sz]),
    "wd_comm_destroy": (_i, [_vp]),
    "wd_synth_plane": (_i, [_vp, _vp, ctypes.POINTER(SynthSpecC), _i, _i, _i]),
    "wd_synth_filter": (_i, [_vp, _vp, ctypes.POINTER(SynthSpecC), _i, _i]),
}

_lib = None


UNITS = ("core", "scan", "queue", "lines", "dense", "ingest")      # csrc/welldup_<unit>.hip -> one object each
OBJ_DIR = os.path.join(HERE, "build_obj")


def _deps(path: str, seen=None) -> set:
    """The file and every csrc/ or include/ file it #includes "by name", transitively."""
    seen = set() if seen is None else seen
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    with open(path, "r") as fh:
        for line in fh:
            line = line.strip()
            if line.startswith('#include "'):
                name = line.split('"')[1]
                for base in (CSRC, INCLUDE):
                    _deps(os.path.join(base, name), seen)
    return seen


def source_build_id() -> str:
    """sha256 over the library's sources (csrc/*, include/welldup.h): what `wd_build_id()` of a library
    built from this tree returns.  Counter profiles and resource tables carry it (tools/pmc_collect.py),
    so that evidence is tied to the code that produced it, not to a kernel's name."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)
                   if f.endswith((".hip", ".inc", ".h")))
    files.append(os.path.join(INCLUDE, "welldup.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def source_unit_ids() -> dict:
    """{unit: sha256[:16] of csrc/welldup_<unit>.hip and everything it includes}: the per-unit part of
    `wd_build_id()`.  A kernel's counter evidence is keyed by the unit it lives in (`unit_of_kernel`)."""
    import hashlib
    out = {}
    for u in UNITS:
        h = hashlib.sha256()
        for f in sorted(_deps(os.path.join(CSRC, "welldup_%s.hip" % u))):
            h.update(os.path.basename(f).encode() + b"\0")
            with open(f, "rb") as fh:
                h.update(fh.read())
        out[u] = h.hexdigest()[:16]
    return out


def parse_build_id(text: str) -> dict:
    """'<all> core=<id> scan=<id> ...' -> {"all": ..., "core": ..., ...}"""
    parts = text.split()
    out = {"all": parts[0] if parts else "unknown"}
    for p in parts[1:]:
        k, _, v = p.partition("=")
        out[k] = v
    return out


def build_ids() -> dict:
    """The loaded library's own account of the sources it was built from (wd_build_id)."""
    return parse_build_id(load().wd_build_id().decode())


def unit_of_kernel(kernel: str) -> str:
    """Translation unit a compare kernel (as wd_last_kernel names it) is compiled in."""
    if kernel.startswith("k_scan_q"):
        return "queue"
    if kernel.startswith("k_scan_lines"):
        return "lines"
    if kernel.startswith(("dense chain", "k_dense")):
        return "dense"
    if kernel.startswith(("k_inflate", "k_cbcl", "k_gather", "k_interleave", "k_scatter")):
        return "ingest"
    return "scan"


def build(force: bool = False, verbose: bool = False, jobs: int = 0) -> str:
    """Compile csrc/welldup_*.hip for gfx950 (one object per unit, in parallel, only the units whose
    sources changed) and link them into the package directory."""
    import concurrent.futures
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    bid = source_build_id()
    uids = source_unit_ids()
    os.makedirs(OBJ_DIR, exist_ok=True)
    stamp = os.path.join(OBJ_DIR, "build_id")
    stale_id = not os.path.exists(stamp) or open(stamp).read().strip() != bid
    todo, objs = [], []
    for u in UNITS:
        src = os.path.join(CSRC, "welldup_%s.hip" % u)
        obj = os.path.join(OBJ_DIR, u + ".o")
        objs.append(obj)
        newest = max(os.path.getmtime(f) for f in _deps(src))
        # (the build id is compiled into the core unit only: four seconds, not a rebuild of every kernel)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < newest or (u == "core" and stale_id):
            todo.append((u, src, obj))
    if not todo and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(o) for o in objs):
        return LIB_PATH

    def compile_one(job):
        u, src, obj = job
        cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-I" + INCLUDE,
               '-DWD_UNIT_ID="%s"' % uids[u], "-c", src, "-o", obj]
        if u == "core":
            cmd.insert(-4, '-DWD_BUILD_ID="%s"' % bid)
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    n = jobs or min(len(todo), os.cpu_count() or 1) or 1
    with concurrent.futures.ThreadPoolExecutor(n) as pool:
        list(pool.map(compile_one, todo))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs + ["-lz"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, "w") as fh:
        fh.write(bid + "\n")
    return LIB_PATH


def load():
    """Load libwelldup.so and bind every prototype; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the scan path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def strerror(code: int) -> str:
    return load().wd_strerror(code).decode()
