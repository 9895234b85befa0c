#!/usr/bin/env python3
"""A/B builds of one translation unit: tools/build_variant.py <tag> <unit> [-DNAME=VALUE ...]
-> well_duplicates_amd/build_variants/libwelldup_<tag>.so (the other units' objects are those of the last
_lib.build()).  Run a probe against it with WELLDUP_LIB=<that path> (well_duplicates_amd/_lib.py)."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from well_duplicates_amd import _lib  # noqa: E402


def main(argv):
    tag, unit, flags = argv[0], argv[1], argv[2:]
    _lib.build()
    out_dir = os.path.join(_lib.HERE, "build_variants")
    os.makedirs(out_dir, exist_ok=True)
    obj = os.path.join(out_dir, "%s_%s.o" % (unit, tag))
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-I" + _lib.INCLUDE,
                           '-DWD_UNIT_ID="variant-%s"' % tag] + flags +
                          ["-c", os.path.join(_lib.CSRC, "welldup_%s.hip" % unit), "-o", obj])
    objs = [obj if u == unit else os.path.join(_lib.OBJ_DIR, u + ".o") for u in _lib.UNITS]
    lib = os.path.join(out_dir, "libwelldup_%s.so" % tag)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-lz"])
    os.remove(obj)
    print(lib)


if __name__ == "__main__":
    main(sys.argv[1:])
