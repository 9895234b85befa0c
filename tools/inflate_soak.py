#!/usr/bin/env python3
"""Soak of the GPU DEFLATE decoder: thousands of damaged .bcl.gz files (bit flips, truncations,
noise runs, header damage; several kinds of payload and compression level), a batch per launch.  Every
file must come back with the verdict Python's gzip module gives the same bytes: the same plane, or an
error - and the process must come back at all.
Usage: inflate_soak.py [rounds] [files per round] [seed] [slim]"""
import gzip
import os
import sys
import tempfile
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from well_duplicates_amd import _lib, synth  # noqa: E402
from well_duplicates_amd.scanner import Scanner  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
per_round = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n = 40001
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
bases = []
for q, lvl in ((7, 6), (7, 1), (39, 6), (2, 9)):
    spec = synth.SynthSpec(seed=q, n_clusters=n, row=200, qual_levels=q)
    bases.append(gzip.compress(synth.bcl_file_bytes(synth.plane_bytes(spec, 1, 1101, 0)), lvl))
runs = (rng.integers(0, 4, n // 40 + 1, dtype=np.uint8).repeat(40)[:n] * 5 + 9).astype(np.uint8)
bases.append(gzip.compress(synth.bcl_file_bytes(runs), 6))
co = zlib.compressobj(6, zlib.DEFLATED, 31, 8, zlib.Z_FIXED)
bases.append(co.compress(synth.bcl_file_bytes(runs)) + co.flush())
stride = (n + 255) // 256 * 256
sc = Scanner(0)
if len(sys.argv) > 4 and sys.argv[4] == "slim":
    # only payloads that expand little, four waves per file: the launches take the decoder's
    # small-window form (k_inflate<4, 256, 4>) whenever the damage leaves the trailers' lengths small
    bases = bases[:3]
    sc.set_option("inflate_waves", 4)
buf = sc.malloc(stride * per_round + 256)
tmp = tempfile.mkdtemp(prefix="wd_soak_")
agree = ok_planes = 0
for r in range(rounds):
    datas, paths = [], []
    for i in range(per_round):
        m = bytearray(bases[int(rng.integers(0, len(bases)))])
        kind = int(rng.integers(0, 5))
        if kind == 0:
            for _ in range(int(rng.integers(1, 5))):
                m[int(rng.integers(10, len(m)))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            m = m[:int(rng.integers(1, len(m)))]
        elif kind == 2:
            at = int(rng.integers(10, len(m)))
            m[at:at + 48] = bytes(rng.integers(0, 256, min(48, len(m) - at), dtype=np.uint8))
        elif kind == 3:
            m[10 + int(rng.integers(0, 60))] ^= 1 << int(rng.integers(0, 8))
        # kind 4: untouched
        p = os.path.join(tmp, "f%d.bcl.gz" % i)
        with open(p, "wb") as fh:
            fh.write(bytes(m))
        datas.append(bytes(m))
        paths.append(p)
    rcs = (_lib.ctypes.c_int * per_round)()
    c_paths = (_lib.ctypes.c_char_p * per_round)(*[p.encode() for p in paths])
    c_dsts = (_lib.ctypes.c_void_p * per_round)(*[buf + i * stride for i in range(per_round)])
    sc._lib.wd_load_bcl_gz_batch(sc._ctx, per_round, c_paths, c_dsts, n, 8, rcs)
    for i in range(per_round):
        try:
            ref = gzip.decompress(datas[i])
            ref_ok = len(ref) >= n + 4 and int.from_bytes(ref[:4], "little") == n
        except (EOFError, zlib.error, gzip.BadGzipFile):
            ref, ref_ok = None, False
        assert (rcs[i] == 0) == ref_ok, (r, i, rcs[i], ref_ok)
        if ref_ok:
            assert sc.d2h(buf + i * stride, n).tobytes() == ref[4:4 + n], (r, i)
            ok_planes += 1
        agree += 1
    print("round %d: %d files, %d verdicts agree so far (%d good planes); on the GPU %d, host %d"
          % (r, per_round, agree, ok_planes, sc.get_option("inflate_files_gpu"), sc.get_option("inflate_files_host")), flush=True)
for f in os.listdir(tmp):
    os.unlink(os.path.join(tmp, f))
os.rmdir(tmp)
print("soak ok")
