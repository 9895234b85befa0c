#!/usr/bin/env python3
"""H2D rate of 16 MB pinned chunks by the NUMA node the pinned memory sits on (set_mempolicy around the
allocation), with the copying thread on either node; the GPU's own node from sysfs."""
import ctypes
import glob
import os
import time

import torch

libc = ctypes.CDLL(None, use_errno=True)
SYS_set_mempolicy = 238            # x86_64
MPOL_DEFAULT, MPOL_BIND = 0, 2


def mempolicy(node):
    if node is None:
        rc = libc.syscall(SYS_set_mempolicy, MPOL_DEFAULT, None, 0)
    else:
        mask = ctypes.c_ulong(1 << node)
        rc = libc.syscall(SYS_set_mempolicy, MPOL_BIND, ctypes.byref(mask), 64)
    if rc != 0:
        print("set_mempolicy failed: errno", ctypes.get_errno())


def cpus_of(node):
    out = []
    for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


torch.cuda.init()
bus = torch.cuda.get_device_properties(0)
print("nodes:", [os.path.basename(n) for n in sorted(glob.glob("/sys/devices/system/node/node[0-9]*"))])
os.system("rocm-smi --showtoponuma 2>/dev/null | grep -i 'numa node'")
MB = 1 << 20
dev = torch.empty(16 * MB, dtype=torch.uint8, device="cuda")
stream = torch.cuda.Stream()
for mem_node in (0, 1):
    mempolicy(mem_node)
    ring = [torch.empty(16 * MB, dtype=torch.uint8).pin_memory() for _ in range(4)]
    for r in ring:
        r.fill_(7)
    mempolicy(None)
    for cpu_node in (0, 1):
        os.sched_setaffinity(0, cpus_of(cpu_node))
        with torch.cuda.stream(stream):
            for _ in range(8):
                dev.copy_(ring[0], non_blocking=True)
            stream.synchronize()
            t0 = time.perf_counter()
            for _ in range(64):
                for k in range(4):
                    dev.copy_(ring[k], non_blocking=True)
            stream.synchronize()
            dt = time.perf_counter() - t0
        print("pinned memory on node %d, issuing thread on node %d: H2D %.1f GB/s" % (mem_node, cpu_node, 256 * 16 * MB / dt / 1e9), flush=True)
    del ring
