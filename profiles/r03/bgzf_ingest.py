"""reader-only timing of the ingest engine on BGZF / gzip / plain FASTQ (no GPU work): where the time of a BGZF source goes.
  python profiles/r03/bgzf_ingest.py [--pairs 4000000]   (env knobs: QM_INGEST_BGZF_THREADS / _DEPTH / _CHUNK)"""
import argparse, os, sys, time, zlib, struct
from multiprocessing import Pool
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)

def blk(c):
    co = zlib.compressobj(1, zlib.DEFLATED, -15); body = co.compress(c) + co.flush(); bs = 12 + 6 + len(body) + 8
    return b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bs - 1) + body + struct.pack("<II", zlib.crc32(c) & 0xffffffff, len(c))

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--pairs", type=int, default=4000000); ap.add_argument("--dir", default="/tmp/bgzf_ingest"); ap.add_argument("--threads", type=int, default=32)
    a = ap.parse_args()
    import rapmap_amd as ra
    from rapmap_amd import synth
    os.makedirs(a.dir, exist_ok=True)
    rng = np.random.default_rng(3)
    p = [os.path.join(a.dir, "r%d.fq" % m) for m in (1, 2)]
    for m, path in enumerate(p):
        seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, a.pairs * 100)]
        synth.write_fastq(path, seq, a.pairs, 100, m + 1)
    with Pool(64) as pool:
        for path in p:
            d = open(path, "rb").read()
            open(path + ".bgzf.gz", "wb").write(b"".join(pool.map(blk, [d[i:i + 65280] for i in range(0, len(d), 65280)], chunksize=64)) + blk(b""))
    L = ra.api.lib()
    def run(sfx, env):
        for k, v in env.items(): os.environ[k] = v
        t = time.time()
        rd = ra.FastxReader(p[0] + sfx, p[1] + sfx, threads=a.threads)
        n = 0
        for b in rd.chunks(1 << 18): n += b.n
        dt = time.time() - t
        rd.close()
        for k in env: del os.environ[k]
        print("%-6s %-60s %.3f s  %.1f M pairs/s  %.2f GB/s" % (sfx or "plain", env, dt, n / dt / 1e6, 2 * a.pairs * 218 / dt / 1e9), flush=True)
    run("", {}); run("", {})
    os.environ["QM_INGEST_DEBUG"] = "1"
    for env in ({}, {"QM_INGEST_BGZF_THREADS": "8"}, {"QM_INGEST_BGZF_THREADS": "32"}, {"QM_INGEST_BGZF_THREADS": "32", "QM_INGEST_BGZF_DEPTH": "16"},
                {"QM_INGEST_BGZF_THREADS": "16", "QM_INGEST_BGZF_JOB": str(1 << 20)}, {"QM_INGEST_BGZF_THREADS": "16", "QM_INGEST_BGZF_JOB": str(64 << 10)},
                {"QM_INGEST_BGZF_THREADS": "64", "QM_INGEST_BGZF_DEPTH": "32", "QM_INGEST_BGZF_CHUNK": str(2 << 20)},
                {"QM_INGEST_BGZF_THREADS": "16", "QM_INGEST_BGZF_DEPTH": "8", "QM_INGEST_BGZF_CHUNK": str(8 << 20)}):
        run(".bgzf.gz", env)
main()
