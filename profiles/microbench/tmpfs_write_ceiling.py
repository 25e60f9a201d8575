#!/usr/bin/env python3
"""How fast can ONE file on tmpfs be written?  (The SAM writer's ceiling: FASTQ -> SAM moves ~1 kB of text per read pair.)
threads x {one shared file via pwrite at disjoint offsets, one shared file via a MAP_SHARED mapping, a file per thread}."""
import mmap
import os
import sys
import tempfile
import threading
import time

GB = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
d = tempfile.mkdtemp(prefix="wceil", dir="/dev/shm")
blk = bytes(bytearray(os.urandom(1 << 20)) * 16)          # 16 MB source block, resident


def run(label, nthr, mode):
    per = int(GB * (1 << 30) / nthr) // len(blk) * len(blk)
    paths = [os.path.join(d, "f%d" % (i if mode == "files" else 0)) for i in range(nthr)]
    fds = [os.open(p, os.O_RDWR | os.O_CREAT | os.O_TRUNC) for p in (paths if mode == "files" else paths[:1])]
    mm = None
    if mode == "mmap":
        os.ftruncate(fds[0], per * nthr)
        mm = mmap.mmap(fds[0], per * nthr, mmap.MAP_SHARED, mmap.PROT_READ | mmap.PROT_WRITE)

    def work(i):
        fd = fds[i] if mode == "files" else fds[0]
        base = 0 if mode == "files" else i * per
        for o in range(0, per, len(blk)):
            if mode == "mmap":
                mm[base + o: base + o + len(blk)] = blk
            else:
                os.pwrite(fd, blk, base + o)
    t = time.time()
    th = [threading.Thread(target=work, args=(i,)) for i in range(nthr)]
    [x.start() for x in th]; [x.join() for x in th]
    dt = time.time() - t
    if mm is not None:
        mm.close()
    for fd in fds:
        os.close(fd)
    for p in set(paths):
        os.unlink(p)
    print("%-44s %2d threads  %6.2f GB/s" % (label, nthr, per * nthr / dt / 1e9), flush=True)


for n in (1, 4, 16):
    run("one file, pwrite at disjoint offsets", n, "pwrite")
for n in (1, 4, 16):
    run("one file, copies into a shared mapping", n, "mmap")
for n in (4, 16):
    run("one file per thread", n, "files")
os.rmdir(d)
