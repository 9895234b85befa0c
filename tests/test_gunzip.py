"""The library's own gunzip (csrc/fast_inflate.inc, behind wd_load_bcl_gz / wd_load_cbcl_tile)
against Python's gzip and against zlib through the same entry point.  Host code only."""
import ctypes
import gzip
import io
import struct
import zlib

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from well_duplicates_amd import _lib, synth

LIB = _lib.load()


def gunzip(data: bytes, cap: int, mode: int):
    src = (ctypes.c_uint8 * max(1, len(data))).from_buffer_copy(data or b"\0")
    dst = (ctypes.c_uint8 * max(1, cap))()
    n = ctypes.c_size_t(0)
    rc = LIB.wd_gunzip(src, len(data), dst, cap, ctypes.byref(n), mode)
    return rc, bytes(dst[:n.value])


def both(data: bytes, want: bytes):
    for mode in (0, 1):
        rc, got = gunzip(data, len(want) + 10, mode)
        assert rc == _lib.OK, mode
        assert got == want, mode


def payloads():
    rng = np.random.default_rng(3)
    spec = synth.SynthSpec(seed=4, n_clusters=300007, row=517, nocall_per_64k=3000)
    plane = struct.pack("<I", spec.n_clusters) + synth.plane_bytes(spec, 1, 1101, 7).tobytes()
    binned = struct.pack("<I", 200000) + (rng.choice([0, 0x1C, 0x5D, 0x9E], size=200000, p=[.01, .2, .3, .49]).astype(np.uint8)
                                          * 1 + rng.integers(0, 4, 200000).astype(np.uint8)).tobytes()
    return {
        "empty": b"",
        "one": b"x",
        "text": b"the quick brown fox jumps over the lazy dog. " * 3000,
        "zeros": bytes(300000),
        "random": rng.integers(0, 256, 200000, dtype=np.uint8).tobytes(),
        "bcl_plane": plane,
        "bcl_binned_quals": binned,
        "runs": b"".join(bytes([int(v)]) * int(n) for v, n in zip(rng.integers(0, 256, 4000), rng.integers(1, 400, 4000))),
        "two_symbols": rng.choice([65, 66], size=100000).astype(np.uint8).tobytes(),
    }


@pytest.mark.parametrize("name", list(payloads()))
@pytest.mark.parametrize("level", [0, 1, 6, 9])
def test_roundtrip(name, level):
    raw = payloads()[name]
    both(gzip.compress(raw, compresslevel=level), raw)


def test_fixed_huffman_and_header_fields():
    raw = b"abcabcabcabc hello hello hello" * 4
    co = zlib.compressobj(9, zlib.DEFLATED, 31, 9, zlib.Z_FIXED)          # fixed codes, gzip wrapper
    both(co.compress(raw) + co.flush(), raw)
    buf = io.BytesIO()
    with gzip.GzipFile(filename="s_1_1101.bcl", mode="wb", fileobj=buf, mtime=1234) as fh:     # FNAME
        fh.write(raw)
    both(buf.getvalue(), raw)
    # FEXTRA + FCOMMENT by hand
    body = zlib.compress(raw, 6)[2:-4]
    head = b"\x1f\x8b\x08" + bytes([4 | 16]) + bytes(6) + struct.pack("<H", 3) + b"xyz" + b"note\0"
    both(head + body + struct.pack("<II", zlib.crc32(raw), len(raw)), raw)
    # ... and with a header CRC: zlib checks it, the library's decoder leaves such files to zlib
    head = b"\x1f\x8b\x08" + bytes([4 | 16 | 2]) + bytes(6) + struct.pack("<H", 3) + b"xyz" + b"note\0"
    data = head + struct.pack("<H", zlib.crc32(head) & 0xFFFF) + body + struct.pack("<II", zlib.crc32(raw), len(raw))
    assert gunzip(data, len(raw) + 10, 0) == (_lib.OK, raw)
    assert gunzip(data, len(raw) + 10, 1)[0] == _lib.ERR_UNSUPPORTED


def test_concatenated_members():
    parts = [b"first member " * 100, b"", b"third " * 5000, bytes(70000)]
    both(b"".join(gzip.compress(p, compresslevel=l) for p, l in zip(parts, (6, 9, 1, 0))), b"".join(parts))


def test_declines_or_rejects_bad_streams():
    raw = payloads()["bcl_plane"][:50000]
    good = gzip.compress(raw, compresslevel=6)
    bad_crc = good[:-8] + struct.pack("<I", zlib.crc32(raw) ^ 1) + good[-4:]
    bad_len = good[:-4] + struct.pack("<I", len(raw) + 1)
    truncated = good[:len(good) // 2]
    garbage_tail = good + b"garbage"
    flipped = bytearray(good)
    flipped[len(good) // 3] ^= 0x55
    for data in (bad_crc, bad_len, truncated, garbage_tail, bytes(flipped), b"not gzip at all", good[:5]):
        rc_z, _ = gunzip(data, len(raw) + 100, 0)
        rc_f, _ = gunzip(data, len(raw) + 100, 1)
        assert rc_z != _lib.OK
        assert rc_f != _lib.OK                      # never "succeeds" where zlib objects
    # output larger than the caller's room
    assert gunzip(good, len(raw) - 1, 0)[0] != _lib.OK
    assert gunzip(good, len(raw) - 1, 1)[0] != _lib.OK


@settings(max_examples=150, deadline=None)
@given(st.binary(max_size=3000), st.integers(0, 9), st.sampled_from([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED,
                                                                       zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE]))
def test_fuzz_small(raw, level, strategy):
    co = zlib.compressobj(level, zlib.DEFLATED, 31, 8, strategy)
    both(co.compress(raw) + co.flush(), raw)


@settings(max_examples=60, deadline=None)
@given(st.binary(min_size=20, max_size=400), st.integers(0, 10**6))
def test_fuzz_corrupt(raw, where):
    """A damaged stream must never crash and never be accepted with other bytes than zlib's."""
    data = bytearray(gzip.compress(raw * 5, compresslevel=6))
    data[where % len(data)] ^= 1 << (where % 8)
    rc_z, out_z = gunzip(bytes(data), 5 * len(raw) + 50, 0)
    rc_f, out_f = gunzip(bytes(data), 5 * len(raw) + 50, 1)
    if rc_f == _lib.OK:
        assert rc_z == _lib.OK and out_f == out_z
