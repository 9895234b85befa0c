"""The GPU DEFLATE decoder's source (csrc/gpu_inflate.inc) run on the CPU: tools/inflate_emu.cpp plays
the lanes of a workgroup with threads, so the control flow of the kernel - sync passes, window caps,
match resolution, every error exit - is checked against zlib here, without a GPU (the GPU tests
compare the kernel itself with Python's gzip: tests/test_gpu_inflate.py)."""
import gzip
import os
import subprocess
import zlib

import numpy as np
import pytest

from well_duplicates_amd import synth

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", params=[(1, 512, 2), (4, 256, 2), (4, 256, 4)],
                ids=["one-wave", "four-waves", "four-waves-small-windows"])
def emu(request, tmp_path_factory):
    waves, span, outdiv = request.param
    exe = str(tmp_path_factory.mktemp("emu") / ("inflate_emu_w%d_d%d" % (waves, outdiv)))
    # CPU build only, under the sanitizers: out-of-bounds writes to the window, stage or match-list
    # memory of a corrupt stream are findings here, not just crashes
    subprocess.check_call(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-std=c++17", "-pthread", "-DEMU_WAVES=%d" % waves, "-DEMU_SPAN=%d" % span,
                           "-DEMU_OUTDIV=%d" % outdiv, os.path.join(REPO, "tools", "inflate_emu.cpp"), "-lz", "-o", exe])
    return exe


def _inputs(tmp):
    rng = np.random.default_rng(3)
    n = 24001
    spec = synth.SynthSpec(seed=3, n_clusters=n, row=250, qual_levels=7)
    bcl = synth.bcl_file_bytes(synth.plane_bytes(spec, 1, 1101, 2, np.arange(n, dtype=np.int64)))
    skew = rng.choice(24, size=20000, p=(lambda p: p / p.sum())(0.5 ** np.arange(1, 25))).astype(np.uint8).tobytes()
    co = zlib.compressobj(6, zlib.DEFLATED, 31, 8, zlib.Z_FIXED)
    files = {
        "bcl_l6": gzip.compress(bcl, 6),
        "bcl_l1": gzip.compress(bcl, 1),
        "skew_l9": gzip.compress(skew, 9),                       # codes longer than the lookup tables' index
        "noise": gzip.compress(rng.integers(0, 256, 20000, dtype=np.uint8).tobytes(), 6),   # stored blocks
        "fixed": co.compress(skew[:9000]) + co.flush(),          # fixed Huffman codes
        "runs": gzip.compress(bytes(rng.integers(0, 4, 600, dtype=np.uint8).repeat(50)), 6),   # overlapping copies
        "tiny": gzip.compress(b"abc"),
        "empty": gzip.compress(b""),
    }
    paths = {}
    for name, data in files.items():
        p = tmp / (name + ".gz")
        p.write_bytes(data)
        paths[name] = str(p)
    return paths


def test_emulated_kernel_agrees_with_zlib(emu, tmp_path):
    paths = _inputs(tmp_path)
    for name, p in paths.items():
        out = subprocess.run([emu, p], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and "MISMATCH" not in out.stdout, (name, out.stdout, out.stderr)
        assert "status 0 " in out.stdout, (name, out.stdout)   # decoded, not declined


def test_emulated_kernel_declines_what_it_cannot_hold(emu, tmp_path):
    p = tmp_path / "zeros.gz"
    p.write_bytes(gzip.compress(bytes(300000), 6))               # one piece of the stream expands > 256-fold
    out = subprocess.run([emu, str(p)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "status 5 " in out.stdout, out.stdout


def test_emulated_kernel_survives_damage(emu, tmp_path):
    """Corrupted streams end with a status (no hang, nothing written past the output) and whatever is
    reported as decoded equals zlib's output."""
    paths = _inputs(tmp_path)
    for name, seed, count in (("fixed", 5, 30), ("bcl_l6", 6, 8)):
        out = subprocess.run([emu, "--fuzz", str(seed), str(count), paths[name]], capture_output=True, text=True,
                             timeout=600)
        assert out.returncode == 0 and "0 disagreements" in out.stdout, (name, out.stdout, out.stderr)
