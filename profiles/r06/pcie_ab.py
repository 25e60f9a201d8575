"""round 6: qm_map_pairs on pageable host buffers + qm_fetch_hits (bench.py's pcie_inclusive leg on its own), three calls; QM_HOST_STAGE=0/1 from the caller.
usage on the GPU box: QM_HOST_STAGE=0 python profiles/r06/pcie_ab.py; QM_HOST_STAGE=1 python profiles/r06/pcie_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
import rapmap_amd as ra

n, L = 10_000_000, 100
cache = os.environ.get("QMAP_BENCH_CACHE", "/tmp/qmap_bench_cache")
idx = bench.build_or_reuse_index(40000, 42, 31, 0, 1, cache, False)
qi = ra.QuasiIndex(idx)
mp = ra.QuasiMapper(qi, 0)
dev = torch.device("cuda:0")
text, starts, lens = bench.load_text_to_gpu(qi, dev)
s1, s2, off = bench.make_reads_gpu(text, starts, lens, n, 43, dev, read_len=L)
hs1, hs2, hoff = s1.cpu().numpy(), s2.cpu().numpy(), off.cpu().numpy()
del s1, s2, text
opts = ra.default_opts()
mp.map_pairs(hs1[: 1000 * L], hoff[:1001], hs2[: 1000 * L], hoff[:1001], opts=opts)
import ctypes as C
from rapmap_amd import api
lib = api.lib()
for i in range(3):
    nh, ctr = C.c_int64(0), api.QmCounters()
    t0 = time.perf_counter()
    rc = lib.qm_map_pairs(mp._h, C.byref(opts), n, hs1.ctypes.data, hoff.ctypes.data, hs2.ctypes.data, hoff.ctypes.data, C.byref(nh), C.byref(ctr))
    t1 = time.perf_counter()
    offs = np.empty(n + 1, dtype=np.int64); hits = np.empty(nh.value, dtype=api.HIT_DTYPE)
    rc2 = lib.qm_fetch_hits(mp._h, offs.ctypes.data, hits.ctypes.data)
    t2 = time.perf_counter()
    dt = t2 - t0
    print("QM_HOST_STAGE=%s call %d: %.1f M pairs/s (%.0f ms = map %.0f + fetch %.0f; rc %d %d, %d hits)" % (
        os.environ.get("QM_HOST_STAGE", "default"), i, n / dt / 1e6, dt * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3, rc, rc2, nh.value), flush=True)
