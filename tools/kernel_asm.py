#!/usr/bin/env python3
"""Disassembly of one kernel of one translation unit's gfx950 code object:

    python tools/kernel_asm.py <unit> <mangled-name regex> [--mem]      (e.g. dense 'k_dense_verifyILb1')

--mem prints only the vector-memory instructions, the waits and the branches (line numbers kept): enough to
see which loads are in flight together."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def main(argv):
    unit, pat = argv[0], re.compile(argv[1])
    mem = "--mem" in argv
    tmp = tempfile.mkdtemp(prefix="wd_asm_")
    try:
        obj = os.path.join(tmp, unit + ".o")
        shutil.copy(os.path.join(REPO, "well_duplicates_amd", "build_obj", unit + ".o"), obj)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", obj], check=True, capture_output=True, cwd=tmp)
        co = [f for f in os.listdir(tmp) if "gfx950" in f][0]
        text = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", os.path.join(tmp, co)], check=True,
                              capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    on, n = False, 0
    for line in text.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            on = bool(pat.search(m.group(1)))
            n = 0
            if on:
                print(line)
            continue
        if not on:
            continue
        n += 1
        body = line.split("//")[0].rstrip()
        if mem and not re.search(r"global_|buffer_|flat_|scratch_|s_waitcnt|s_cbranch|s_branch|s_endpgm|s_barrier", body):
            continue
        print("%5d %s" % (n, body.strip()))


if __name__ == "__main__":
    main(sys.argv[1:])
