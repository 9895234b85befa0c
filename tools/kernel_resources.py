#!/usr/bin/env python3
"""Per-kernel register / spill / LDS figures from the gfx950 code object inside libwelldup.so.

    python tools/kernel_resources.py [pattern ...] [--json FILE]

The numbers are the compiler's own (the AMDGPU metadata note of the code object: llvm-readelf
--notes), so they describe exactly the binary that ships - no GPU needed.  Patterns are substrings
of the demangled kernel names (default: the scan kernels)."""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
FIELDS = [("vgpr", ".vgpr_count"), ("agpr", ".agpr_count"), ("sgpr", ".sgpr_count"),
          ("sgpr_spills", ".sgpr_spill_count"), ("vgpr_spills", ".vgpr_spill_count"),
          ("lds_bytes", ".group_segment_fixed_size"), ("scratch_bytes", ".private_segment_fixed_size"),
          ("max_flat_workgroup_size", ".max_flat_workgroup_size")]


def kernels(lib=None):
    lib = lib or os.path.join(REPO, "well_duplicates_amd", "libwelldup.so")
    tmp = tempfile.mkdtemp(prefix="wd_co_")
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], check=True, capture_output=True, cwd=tmp)
        cos = [f for f in os.listdir(tmp) if "gfx950" in f]
        if not cos:
            raise SystemExit("no gfx950 code object in %s" % lib)
        # one code object per translation unit of the library (csrc/welldup_*.hip)
        notes = "\n".join(subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, co)],
                                         check=True, capture_output=True, text=True).stdout for co in sorted(cos))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {}
    # the metadata is YAML: one "- .agpr_count: ..." block per kernel under amdhsa.kernels
    for block in re.split(r"\n\s+- (?=\.)", notes):
        m = re.search(r"\.name:\s+(\S+)", block)
        if not m or ".sgpr_count" not in block:
            continue
        rec = {}
        for key, field in FIELDS:
            v = re.search(r"%s:\s+(\d+)" % re.escape(field), block)
            rec[key] = int(v.group(1)) if v else None
        out[m.group(1)] = rec
    names = list(out)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return {d.replace("(ScanArgs)", "").replace("(LineArgs)", "").replace("(DenseArgs)", "")
             .replace("(anonymous namespace)::", ""): out[n] for n, d in zip(names, dem)}


def build_id(lib=None):
    """wd_build_id() of the library the figures were read from (no GPU needed: a plain string getter)."""
    import ctypes
    h = ctypes.CDLL(lib or os.path.join(REPO, "well_duplicates_amd", "libwelldup.so"))
    h.wd_build_id.restype = ctypes.c_char_p
    return h.wd_build_id().decode()           # "<all> core=<id> scan=<id> ..." (include/welldup.h)


def main(argv):
    js = None
    if "--json" in argv:
        i = argv.index("--json")
        js = argv[i + 1]
        argv = argv[:i] + argv[i + 2:]
    pats = argv or ["k_scan_q", "k_scan_lines", "k_dense_"]
    ks = kernels()
    sel = {k: v for k, v in sorted(ks.items()) if any(p in k for p in pats)}
    print("%-58s %5s %5s %6s %6s %7s %8s" % ("kernel", "vgpr", "sgpr", "s.spill", "v.spill", "lds", "scratch"))
    for k, v in sel.items():
        print("%-58s %5s %5s %6s %6s %7s %8s" % (k[:58], v["vgpr"], v["sgpr"], v["sgpr_spills"], v["vgpr_spills"],
                                                   v["lds_bytes"], v["scratch_bytes"]))
    if js:
        with open(js, "w") as fh:
            json.dump({"build_id": build_id(), "kernels": sel}, fh, indent=1, sort_keys=True)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
