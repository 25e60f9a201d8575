"""FASTQ.gz -> hits through the pipelined stream (builder-run on the GPU box), round 5: an ORDINARY gzip file -- one deflate stream,
what `gzip -6` writes, what real callers have -- inflated by several threads (rapmap_amd/csrc/qm_pgz.h), next to the same two
files as plain FASTQ and through the single zlib stream the reference reads them with (src/FastxParser.cpp:229-328;
QM_INGEST_NO_PGZ=1).  The index is bench.py's cached config-2 index.

  python profiles/r05/e2e_gz.py [--pairs 4000000] [--threads 48] [--dir /tmp/e2e_gz] [--level 6]"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=4000000)
    ap.add_argument("--threads", type=int, default=48)
    ap.add_argument("--dir", default="/tmp/e2e_gz")
    ap.add_argument("--level", type=int, default=6)
    a = ap.parse_args()
    import bench
    import rapmap_amd as ra
    os.makedirs(a.dir, exist_ok=True)
    log = lambda *x: print(*x, file=sys.stderr, flush=True)
    ra.reserve_stream_memory(768 << 20)
    cache = os.environ.get("QMAP_BENCH_CACHE", "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
    idx = bench.build_or_reuse_index(40000, 42, 31, 0, 1, cache)
    qi = ra.QuasiIndex(idx)
    from rapmap_amd import synth
    import torch
    dev = torch.device("cuda:0")
    text, starts, lens = bench.load_text_to_gpu(qi, dev)
    s1, s2, off = bench.make_reads_gpu(text, starts, lens, a.pairs, 43, dev, read_len=100)
    p = [os.path.join(a.dir, "r_%d.fq" % m) for m in (1, 2)]
    t = time.time()
    for m, (path, s) in enumerate(zip(p, (s1, s2))):
        synth.write_fastq(path, s.cpu().numpy()[: a.pairs * 100], a.pairs, 100, m + 1)
    del text, s1, s2
    log("fastq written %.1fs" % (time.time() - t))
    t = time.time()
    ps = [subprocess.Popen("gzip -%d -c %s > %s.gz" % (a.level, x, x), shell=True) for x in p]
    for q in ps:
        assert q.wait() == 0
    log("gzip -%d: %.1fs" % (a.level, time.time() - t))
    out = {"pairs": a.pairs, "threads": a.threads, "gzip_level": a.level,
           "bytes": {k: sum(os.path.getsize(x + sfx) for x in p) for k, sfx in (("plain", ""), ("gzip", ".gz"))}}
    mp = ra.QuasiMapper(qi, 0)          # keeps the device's index replica alive: the streams' contexts share it (as in bench.py, as in the CLI)
    os.environ["QM_INGEST_PIN"] = "1"
    want = None
    runs = [("plain", "", {}), ("gzip_parallel", ".gz", {}), ("gzip_parallel", ".gz", {}),
            ("gzip_parallel_8_threads_per_file", ".gz", {"QM_INGEST_PGZ_THREADS": "8"}), ("gzip_parallel_32_threads_per_file", ".gz", {"QM_INGEST_PGZ_THREADS": "32"}),
            ("gzip_one_zlib_stream", ".gz", {"QM_INGEST_NO_PGZ": "1"})]
    for kind, sfx, env in runs:
        for k_, v_ in env.items():
            os.environ[k_] = v_
        os.environ["QM_INGEST_DEBUG"] = "1"
        t = time.time()
        st = ra.MappedStream(qi, p[0] + sfx, p[1] + sfx, device=0, batch_units=1 << 18, threads=a.threads, names=False)
        n = 0; hits = 0
        for b in st:
            n += b.n; hits += b.n_hits
        dt = time.time() - t
        st.close()
        for k_ in env:
            del os.environ[k_]
        assert n == a.pairs
        if want is None:
            want = hits
        assert hits == want, (kind, hits, want)
        out.setdefault(kind, []).append({"seconds": round(dt, 3), "M_pairs_per_s": round(n / dt / 1e6, 2)})
        log(kind, out[kind][-1])
    out["hits"] = want
    print(json.dumps(out))


if __name__ == "__main__":
    main()
