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
        import ctypes as C
        from rapmap_amd import api as _api
        L = _api.lib()
        nh, ctr = C.c_int64(0), _api.QmCounters()
        opts = ra.default_opts()
        t0 = time.perf_counter()
        _api._check(L.qm_map_pairs(mp._h, C.byref(opts), pairs, a1.ctypes.data, ao.ctypes.data, a2.ctypes.data, ao.ctypes.data,
                                     C.byref(nh), C.byref(ctr)))
        tm = time.perf_counter()
        res = mp._finish(pairs, nh, ctr)                      # allocates the result arrays and calls qm_fetch_hits
        t1 = time.perf_counter()
        rate = pairs / (t1 - t0) / 1e6
        tag = "warmup" if r == 0 else "rep %d" % r
        print("[pcie] %s: %.1f ms host->hits on host (%.1f M pairs/s) = qm_map_pairs %.1f ms (stage A span %.1f ms, device total %.1f ms) + "
              "allocate and qm_fetch_hits %.1f ms, %d hits, pinned=%s"
              % (tag, (t1 - t0) * 1e3, rate, (tm - t0) * 1e3, res.map_kernel_ms, res.total_ms, (t1 - tm) * 1e3, res.n_hits, pinned), flush=True)
        if r > 0:
            best = max(best, rate)
    print("[pcie] best %.1f M pairs/s (H2D of %.2f GB reads + map + D2H of %.2f GB hits)"
          % (best, (a1.nbytes + a2.nbytes + 2 * ao.nbytes) / 1e9, (res.hits.nbytes + res.hit_offsets.nbytes) / 1e9))


if __name__ == "__main__":
    main()
