"""FASTQ.gz -> hits through the pipelined stream, three encodings of the same two files (builder-run on the GPU box):
plain FASTQ, ordinary gzip (one deflate stream per file: one zlib thread each, as the reference's kseq reader has it,
src/FastxParser.cpp:229-328), BGZF (bgzip / bcl2fastq: independent gzip members inflated by helper threads side by side,
rapmap_amd/csrc/qm_ingest.cpp BgzfReader).  The index is bench.py's cached config-2 index.

  python profiles/r03/e2e_gz.py [--pairs 4000000] [--threads 32] [--dir /tmp/e2e_gz]"""
import argparse
import json
import os
import sys
import time
import zlib
from multiprocessing import Pool

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def _bgzf_block(c):
    import struct
    co = zlib.compressobj(1, zlib.DEFLATED, -15)
    body = co.compress(c) + co.flush()
    bsize = 12 + 6 + len(body) + 8
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1) + body +
            struct.pack("<II", zlib.crc32(c) & 0xffffffff, len(c) & 0xffffffff))


def _gz_whole(args):
    src, dst = args
    import gzip
    with open(src, "rb") as f, gzip.open(dst, "wb", compresslevel=1) as g:
        while True:
            b = f.read(1 << 24)
            if not b:
                break
            g.write(b)
    return dst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=4000000)
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--dir", default="/tmp/e2e_gz")
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
    with Pool(min(64, os.cpu_count() or 8)) as pool:
        for path in p:
            data = open(path, "rb").read()
            blocks = pool.map(_bgzf_block, [data[i:i + 65280] for i in range(0, len(data), 65280)], chunksize=64)
            with open(path + ".bgzf.gz", "wb") as f:
                f.write(b"".join(blocks) + _bgzf_block(b""))
        pool.map(_gz_whole, [(x, x + ".gz") for x in p])
    log("compressed %.1fs" % (time.time() - t))
    out = {"pairs": a.pairs, "threads": a.threads, "bytes": {k: sum(os.path.getsize(x + sfx) for x in p) for k, sfx in (("plain", ""), ("gzip", ".gz"), ("bgzf", ".bgzf.gz"))}}
    mp = ra.QuasiMapper(qi, 0)          # keeps the device's index replica alive: the streams' contexts share it (as in bench.py, as in the CLI)
    want = None
    for kind, sfx in (("plain", ""), ("bgzf", ".bgzf.gz"), ("gzip", ".gz"), ("plain", ""), ("bgzf", ".bgzf.gz")):
        t = time.time()
        st = ra.MappedStream(qi, p[0] + sfx, p[1] + sfx, device=0, batch_units=1 << 18, threads=a.threads, names=False)
        n = 0; hits = 0; ctr = None
        for b in st:
            n += b.n; hits += b.n_hits
        dt = time.time() - t
        st.close()
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
