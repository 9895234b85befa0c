#!/usr/bin/env python3
"""What costs the ingest's chunk copies their rate?  H2D copies of 16 MB pinned chunks on one stream
(a) alone, (b) beside threads that copy memory that has nothing to do with them, (c) beside threads
that pread a page-cached file into OTHER pinned chunks, as the reader threads of wd_load_tile_files_batch do."""
import os
import sys
import tempfile
import threading
import time

import numpy as np
import torch

MB = 1 << 20
chunk = 16 * MB
ring = [torch.empty(chunk, dtype=torch.uint8).pin_memory() for _ in range(8)]
dev = torch.empty(chunk, dtype=torch.uint8, device="cuda")
stream = torch.cuda.Stream()
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 16
tmp = tempfile.NamedTemporaryFile(prefix="wd_h2d_", delete=False)
tmp.write(os.urandom(64 * MB) * 4)                      # 256 MB, stays in the page cache
tmp.close()


def copies(seconds):
    n = 0
    with torch.cuda.stream(stream):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for k in range(4):
                dev.copy_(ring[k], non_blocking=True)
            stream.synchronize()
            n += 4
        dt = time.perf_counter() - t0
    return n * chunk / dt / 1e9


def run(name, worker):
    stop = threading.Event()
    done = [0] * threads
    ts = [threading.Thread(target=worker, args=(i, stop, done)) for i in range(threads if worker else 0)]
    for t in ts:
        t.start()
    time.sleep(0.1)
    t0 = time.perf_counter()
    rate = copies(1.5)
    dt = time.perf_counter() - t0
    stop.set()
    for t in ts:
        t.join()
    print("%-46s H2D %.1f GB/s   (the threads moved %.1f GB/s)" % (name, rate, sum(done) / dt / 1e9), flush=True)


def memcpy_worker(i, stop, done):
    a = np.ones(32 * MB, np.uint8)
    b = np.empty_like(a)
    while not stop.is_set():
        np.copyto(b, a)
        done[i] += a.nbytes


def pread_worker(i, stop, done):
    fd = os.open(tmp.name, os.O_RDONLY)
    view = memoryview(ring[4 + i % 4].numpy())          # chunks the DMA is not reading
    part = chunk // 4
    off = 0
    while not stop.is_set():
        got = os.preadv(fd, [view[(i // 4 % 4) * part:(i // 4 % 4 + 1) * part]], off % (192 * MB))
        done[i] += got
        off += part
    os.close(fd)


copies(0.5)
run("alone", None)
run("beside %d threads copying unrelated memory" % threads, memcpy_worker)
run("beside %d threads pread-ing into pinned chunks" % threads, pread_worker)
run("alone again", None)
os.unlink(tmp.name)
