"""Does -s gain from two batches in flight on one GPU?  The collector is bound by the scalar unit and by waiting, the ksw2 kernels
by the VALU: two contexts (own stream each) map the two halves of the 10 M-pair batch at the same time; compare with one context
mapping the whole batch.  python profiles/r04/sel_overlap_probe.py   (GPU box; prints M pairs/s)"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import rapmap_amd as ra

def main():
    dev = torch.device("cuda:0"); torch.cuda.init()
    idx = bench.build_or_reuse_index(40000, 42, 31, 0, 1, os.environ.get("QMAP_BENCH_CACHE", "/dev/shm"), False)
    qi = ra.QuasiIndex(idx)
    mps = [ra.QuasiMapper(qi, 0) for _ in range(3)]
    text, starts, lens = bench.load_text_to_gpu(qi, dev)
    n, L = 10_000_000, 100
    s1, s2, off = bench.make_reads_gpu(text, starts, lens, n, 43, dev, read_len=L)
    del text; torch.cuda.synchronize()
    for sel in (1, 0):
        opts = ra.default_opts(sel_aln=1) if sel else ra.default_opts()
        def run(mp, u0, u1):
            return mp.map_device(u1 - u0, s1.data_ptr(), off.data_ptr() + 8 * u0, s2.data_ptr(), off.data_ptr() + 8 * u0, L, opts=opts, fetch=False)
        for parts in (1, 2, 3):
            bounds = [n * i // parts for i in range(parts + 1)]
            def step():
                th = [threading.Thread(target=run, args=(mps[i], bounds[i], bounds[i + 1])) for i in range(parts)]
                for t in th: t.start()
                for t in th: t.join()
            step(); step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            K = 4
            for _ in range(K): step()
            torch.cuda.synchronize(); el = (time.perf_counter() - t0) / K
            print("sel=%d contexts=%d: %.1f ms per 10 M pairs, %.1f M pairs/s" % (sel, parts, el * 1e3, n / el / 1e6), flush=True)

main()
