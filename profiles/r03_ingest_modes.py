"""Ingest engine on the GPU box's host: mmap vs pread chunks, first pass over freshly written files vs second pass, worker
counts; reader alone (malloc'd slots) and the whole stream (pinned slots + mapping).  python profiles/r03_ingest_modes.py [pairs]"""
import json, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import rapmap_amd as ra
from rapmap_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
L = 100
dev = torch.device("cuda", 0)
idx_dir = bench.build_or_reuse_index(40000, 42, 31, 0, 1, "/dev/shm")
qi = ra.QuasiIndex(idx_dir)
text, starts, lens = bench.load_text_to_gpu(qi, dev)
s1, s2, off = bench.make_reads_gpu(text, starts, lens, n, 43, dev, read_len=L)
h1 = s1[: n * L].cpu().numpy(); h2 = s2[: n * L].cpu().numpy()
del s1, s2, text
torch.cuda.empty_cache()
keep = ra.QuasiMapper(qi, 0)
d = "/dev/shm/qmap_e2e"; os.makedirs(d, exist_ok=True)
serial = [0]

def fresh():
    serial[0] += 1
    f1, f2 = d + "/r1_%d.fq" % serial[0], d + "/r2_%d.fq" % serial[0]
    synth.write_fastq(f1, h1, n, L, 1); synth.write_fastq(f2, h2, n, L, 2)
    return f1, f2

def reader(f1, f2, thr, tag):
    t = time.perf_counter(); rd = ra.FastxReader(f1, f2, threads=thr); tot = 0
    for b in rd.chunks(1 << 18):
        tot += b.n
    rd.close(); dt = time.perf_counter() - t
    print(json.dumps({"what": "reader", "tag": tag, "threads": thr, "M_pairs_s": round(tot / dt / 1e6, 2), "GB_s": round(2 * os.path.getsize(f1) / dt / 1e9, 2)}), flush=True)

def stream(f1, f2, thr, tag, names=True):
    t = time.perf_counter()
    st = ra.MappedStream(qi, f1, f2, device=0, batch_units=1 << 18, threads=thr, names=names)
    nh = 0
    for b in st:
        nh += b.n_hits
    dt = time.perf_counter() - t
    ss = st.stats(); st.close()
    print(json.dumps({"what": "stream", "tag": tag, "threads": thr, "M_pairs_s": round(n / dt / 1e6, 2), "s": round(dt, 4),
                      **{k: round(v, 4) for k, v in ss.items() if k in ("read_s", "map_s", "open_s", "first_batch_s", "parse_cpu_s", "copy_cpu_s", "caller_wait_s")}}), flush=True)

for mode in ("0", "1"):
    os.environ["QM_INGEST_PREAD"] = mode
    for thr in (32, 64):
        f1, f2 = fresh()
        stream(f1, f2, thr, "pread=%s first pass" % mode)
        stream(f1, f2, thr, "pread=%s second pass" % mode)
        stream(f1, f2, thr, "pread=%s third pass" % mode)
        os.remove(f1); os.remove(f2)
    f1, f2 = fresh()
    reader(f1, f2, 32, "pread=%s first pass" % mode)
    reader(f1, f2, 32, "pread=%s second pass" % mode)
    reader(f1, f2, 16, "pread=%s" % mode)
    reader(f1, f2, 48, "pread=%s" % mode)
    reader(f1, f2, 64, "pread=%s" % mode)
    os.remove(f1); os.remove(f2)
# chunk size
os.environ["QM_INGEST_PREAD"] = "1"
f1, f2 = fresh()
stream(f1, f2, 32, "warmup")
for ch in (1 << 19, 1 << 20, 1 << 22):
    os.environ["QM_INGEST_CHUNK"] = str(ch)
    stream(f1, f2, 32, "pread chunk %d" % ch)
    stream(f1, f2, 48, "pread chunk %d" % ch)
os.remove(f1); os.remove(f2)
