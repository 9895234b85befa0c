"""The C-ABI shared library: builds for gfx950, loads, and exports every symbol the header
declares (no compute calls - this runs without a GPU)."""
import os
import re

import pytest

from well_duplicates_amd import _lib

HEADER = os.path.join(_lib.INCLUDE, "welldup.h")


@pytest.fixture(scope="module")
def lib():
    _lib.build()
    return _lib.load()


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wd_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(lib):
    syms = declared_symbols()
    assert len(syms) >= 25
    assert sorted(_lib.PROTOTYPES) == syms
    for s in syms:
        assert getattr(lib, s) is not None


def test_error_strings_and_version(lib):
    assert lib.wd_version() >= 100
    assert lib.wd_strerror(0) == b"ok"
    for code in range(-8, 0):
        assert lib.wd_strerror(code) not in (b"", b"unknown error")
    assert lib.wd_strerror(-99) == b"unknown error"


def test_build_id_is_the_hash_of_the_sources(lib):
    """Counter profiles and resource tables carry wd_build_id(): it must name THIS tree's sources."""
    ids = _lib.build_ids()
    assert re.fullmatch(r"[0-9a-f]{16}", ids["all"]), ids
    assert ids["all"] == _lib.source_build_id()
    units = _lib.source_unit_ids()
    assert sorted(units) == sorted(_lib.UNITS)
    for u, h in units.items():
        assert ids[u] == h, (u, ids)
    assert _lib.unit_of_kernel("k_scan_q<true, 2, 0, 1>, targets sorted by centre") == "queue"
    assert _lib.unit_of_kernel("k_scan_lines<true, 5, -1>") == "lines"
    assert _lib.unit_of_kernel("dense chain v5, equality") == "dense"
    assert _lib.unit_of_kernel("k_scan<HamState, true, 4, 4>") == "scan"


def test_header_constants_match_binding():
    text = open(HEADER).read()
    consts = dict(re.findall(r"#define (WD_[A-Z_]+) \(?(-?\d+)\)?", text))
    assert int(consts["WD_ERR_INDEX"]) == _lib.ERR_INDEX
    assert int(consts["WD_ERR_EMPTY_LEVEL"]) == _lib.ERR_EMPTY_LEVEL
    assert int(consts["WD_MODE_LEVENSHTEIN"]) == _lib.MODE_LEVENSHTEIN
    assert int(consts["WD_MAX_LEVELS"]) == _lib.MAX_LEVELS
    assert int(consts["WD_UNIQUE_ID_BYTES"]) == _lib.UNIQUE_ID_BYTES


def test_no_gpu_fails_loudly(lib):
    """Without a GPU the product path refuses to run - there is no CPU fallback."""
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present")
    assert not lib.wd_create(0)
    assert lib.wd_create_status() != 0
    from well_duplicates_amd.scanner import Scanner
    with pytest.raises(RuntimeError):
        Scanner(0)


def test_product_never_imports_oracle():
    root = os.path.dirname(_lib.HERE)
    for dirpath, _, files in os.walk(_lib.HERE):
        for f in files:
            if f.endswith((".py", ".hip", ".inc", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "libwelldup_oracle" not in src, f
