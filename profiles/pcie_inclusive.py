#!/usr/bin/env python3
"""PCIe-inclusive rate of the C-ABI host entry point: qm_map_pairs (host buffers in, H2D + map) followed by
qm_fetch_hits (D2H), on the bench workload (config 2).  Usage: python profiles/pcie_inclusive.py [pairs] [reps]
Prints one line per repetition and the best; never part of bench.py's `value`."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import rapmap_amd as ra  # noqa: E402


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    pinned = os.environ.get("QM_PINNED", "0") == "1"
    dev = torch.device("cuda", 0)
    idx = bench.build_or_reuse_index(40000, 42, 31, 0, 1, "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
    qi = ra.QuasiIndex(idx)
    mp = ra.QuasiMapper(qi, 0)
    text, starts, lens = bench.load_text_to_gpu(qi, dev)
    s1, s2, off = bench.make_reads_gpu(text, starts, lens, pairs, 43, dev, read_len=100)
    h1 = s1.cpu(); h2 = s2.cpu(); ho = off.cpu()
    if pinned:
        h1 = h1.pin_memory(); h2 = h2.pin_memory(); ho = ho.pin_memory()
    a1, a2, ao = h1.numpy(), h2.numpy(), ho.numpy()
    del s1, s2, off
    best = 0.0
    for r in range(reps + 1):
        t0 = time.perf_counter()
        res = mp.map_pairs(a1, ao, a2, ao)
        t1 = time.perf_counter()
        rate = pairs / (t1 - t0) / 1e6
        tag = "warmup" if r == 0 else "rep %d" % r
        print("[pcie] %s: %.1f ms host->hits on host (%.1f M pairs/s), map kernel %.1f ms, device total %.1f ms, %d hits, pinned=%s"
              % (tag, (t1 - t0) * 1e3, rate, res.map_kernel_ms, res.total_ms, res.n_hits, pinned), flush=True)
        if r > 0:
            best = max(best, rate)
    print("[pcie] best %.1f M pairs/s (H2D of %.2f GB reads + map + D2H of %.2f GB hits)"
          % (best, (a1.nbytes + a2.nbytes + 2 * ao.nbytes) / 1e9, (res.hits.nbytes + res.hit_offsets.nbytes) / 1e9))


if __name__ == "__main__":
    main()
