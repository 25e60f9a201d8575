"""Ingest engine scaling hypotheses on the GPU box's host (reader alone, 20 M pairs, files in /tmp page cache):
QM_INGEST_PREAD (no file mapping), QM_INGEST_CHUNK (task rate), CPU affinity (one socket).  python profiles/r04/ingest_hyp.py"""
import json, os, subprocess, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import numpy as np
n = 20_000_000; L = 100
d = "/tmp/qmap_e2e_hyp"; os.makedirs(d, exist_ok=True)
f1, f2 = d + "/r1.fq", d + "/r2.fq"
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import rapmap_amd as ra
    thr = int(sys.argv[2])
    best = 0
    for rep in range(3):
        t = time.perf_counter(); rd = ra.FastxReader(f1, f2, threads=thr); tot = 0
        for b in rd.chunks(1 << 18):
            tot += b.n
        rd.close(); dt = time.perf_counter() - t
        best = max(best, tot / dt / 1e6)
    print(json.dumps({"threads": thr, "M_pairs_s_best_of_3": round(best, 2), "env": {k: v for k, v in os.environ.items() if k.startswith("QM_INGEST")}, "affinity": len(os.sched_getaffinity(0))}), flush=True)
    sys.exit(0)
from rapmap_amd import synth
rng = np.random.default_rng(1)
seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n * L)
synth.write_fastq(f1, seq, n, L, 1); synth.write_fastq(f2, seq, n, L, 2)
del seq
def run(thr, env=None, cpus=None):
    e = dict(os.environ); e.update(env or {})
    cmd = [sys.executable, os.path.abspath(__file__), "child", str(thr)]
    if cpus:
        cmd = ["taskset", "-c", cpus] + cmd
    p = subprocess.run(cmd, env=e, capture_output=True, text=True)
    print((p.stdout.strip().splitlines() or [p.stderr[-300:]])[-1], flush=True)
mode = sys.argv[1] if len(sys.argv) > 1 else "all"
if mode == "after":                                   # after the chain wake-ups / adaptive chunk size of round 4
    for thr in (8, 16, 24, 32, 48, 64, 96, 128):
        run(thr)
    for thr in (16, 32, 64):
        run(thr, {"QM_INGEST_PIN": "0"})
    for thr in (32, 64):
        run(thr, {"QM_INGEST_COPY_RUN": "8192"})
    for thr in (32, 64):
        run(thr, {"QM_INGEST_COPY_RUN": "65536"})
    for thr in (32, 64):
        run(thr, {"QM_INGEST_CHUNK": str(2 << 20)})
    for thr in (32, 64):
        run(thr, {"QM_INGEST_CHUNK": str(16 << 20)})
elif mode == "affinity":
    for thr in (16, 24, 32, 48):
        run(thr, cpus="0-63")                         # one socket's physical cores
    for thr in (32, 64):
        run(thr, cpus="0-127")                        # both sockets' physical cores, no SMT siblings
    for thr in (16, 32):
        run(thr, cpus="0-15,64-79")                   # 16 + 16 cores, one thread per core when thr = 32
    for thr in (16, 32):
        run(thr, cpus="0-31")
else:
    for thr in (8, 16, 32, 64):
        run(thr)
    for thr in (16, 32, 64):
        run(thr, {"QM_INGEST_PREAD": "1"})
    for thr in (16, 32, 64):
        run(thr, {"QM_INGEST_CHUNK": str(8 << 20)})
    for thr in (16, 32, 64):
        run(thr, {"QM_INGEST_CHUNK": str(512 << 10)})
    for thr in (16, 32, 64):
        run(thr, cpus="0-63")
    for thr in (32, 64):
        run(thr, cpus="0-31,64-95")
    for thr in (16, 32, 64):
        run(thr, {"QM_INGEST_PREAD": "1", "QM_INGEST_CHUNK": str(8 << 20)})
os.remove(f1); os.remove(f2)
