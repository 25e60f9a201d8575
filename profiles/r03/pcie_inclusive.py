"""qm_map_pairs on pageable host buffers + qm_fetch_hits (bench.py's pcie_inclusive leg on its own) and where its time goes:
the map call, allocating the result arrays in Python, the fetch into fresh / into touched memory.  (QM_COPY_THREADS belonged to
an experiment -- the caller's bytes staged into pinned buffers by several host threads instead of hipMemcpyAsync from pageable
memory -- that changed nothing: 135 ms either way, the upload hides under the kernels.)  python profiles/r03/pcie_inclusive.py"""
import json, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
import rapmap_amd as ra
n, L = 10_000_000, 100
dev = torch.device("cuda", 0)
idx_dir = bench.build_or_reuse_index(40000, 42, 31, 0, 1, "/dev/shm")
qi = ra.QuasiIndex(idx_dir)
text, starts, lens = bench.load_text_to_gpu(qi, dev)
s1, s2, off = bench.make_reads_gpu(text, starts, lens, n, 43, dev, read_len=L)
hs1 = s1.cpu().numpy(); hs2 = s2.cpu().numpy(); hoff = off.cpu().numpy()
del s1, s2, text
for thr in ("8", "16", "4"):
    os.environ["QM_COPY_THREADS"] = thr
    mp = ra.QuasiMapper(qi, 0)
    mp.map_pairs(hs1[: 1000 * L], hoff[:1001], hs2[: 1000 * L], hoff[:1001])
    for rep in range(3):
        t = time.perf_counter(); r = mp.map_pairs(hs1, hoff, hs2, hoff); dt = time.perf_counter() - t
        print(json.dumps({"copy_threads": int(thr), "rep": rep, "M_pairs_s": round(n / dt / 1e6, 2), "ms": round(dt * 1e3, 1), "hits": int(r.n_hits), "kernel_ms": round(r.map_kernel_ms, 2)}), flush=True)
    mp.close()

import ctypes as C
from rapmap_amd.api import lib, QmCounters, default_opts, HIT_DTYPE
L_ = lib(); mp = ra.QuasiMapper(qi, 0); opts = default_opts()
mp.map_pairs(hs1[: 1000 * L], hoff[:1001], hs2[: 1000 * L], hoff[:1001])
for rep in range(3):
    nh, ctr = C.c_int64(0), QmCounters()
    t0 = time.perf_counter()
    rc = L_.qm_map_pairs(mp._h, C.byref(opts), n, hs1.ctypes.data, hoff.ctypes.data, hs2.ctypes.data, hoff.ctypes.data, C.byref(nh), C.byref(ctr))
    t1 = time.perf_counter()
    ho = np.zeros(n + 1, dtype=np.int64); hh = np.zeros(nh.value, dtype=HIT_DTYPE)
    t2 = time.perf_counter()
    L_.qm_fetch_hits(mp._h, ho.ctypes.data, hh.ctypes.data)
    t3 = time.perf_counter()
    L_.qm_fetch_hits(mp._h, ho.ctypes.data, hh.ctypes.data)
    t4 = time.perf_counter()
    print(json.dumps({"map_ms": round((t1 - t0) * 1e3, 1), "alloc_ms": round((t2 - t1) * 1e3, 1), "fetch_fresh_ms": round((t3 - t2) * 1e3, 1), "fetch_warm_ms": round((t4 - t3) * 1e3, 1)}), flush=True)
