"""Where does the ingest engine stop scaling on the GPU box's two-socket host (256 hardware threads)?  FASTQ files of N pairs
written with the default memory policy (page cache on the writer's node) and with MPOL_INTERLEAVE (spread over both sockets),
read by the engine alone (reader: malloc'd slots) and by the whole stream (pinned slots + mapping) at several worker counts.
python profiles/r04/ingest_numa.py [pairs]"""
import json, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import rapmap_amd as ra
from rapmap_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
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
ra.reserve_stream_memory(1536 << 20)
d = "/tmp/qmap_e2e_numa"; os.makedirs(d, exist_ok=True)
os.system("lscpu | grep -i 'numa\\|socket\\|^CPU(s)' 1>&2")

def write(tag, interleave):
    f1, f2 = d + "/r1_%s.fq" % tag, d + "/r2_%s.fq" % tag
    if interleave:
        assert bench.interleave_memory(True)
    t = time.time()
    synth.write_fastq(f1, h1, n, L, 1); synth.write_fastq(f2, h2, n, L, 2)
    if interleave:
        bench.interleave_memory(False)
    print(json.dumps({"what": "write", "tag": tag, "s": round(time.time() - t, 1)}), flush=True)
    return f1, f2

def reader(f1, f2, thr, tag):
    t = time.perf_counter(); rd = ra.FastxReader(f1, f2, threads=thr); tot = 0
    for b in rd.chunks(1 << 18):
        tot += b.n
    rd.close(); dt = time.perf_counter() - t
    print(json.dumps({"what": "reader", "files": tag, "threads": thr, "M_pairs_s": round(tot / dt / 1e6, 2), "GB_s": round(2 * os.path.getsize(f1) / dt / 1e9, 2)}), flush=True)

def stream(f1, f2, thr, tag, names=False):
    t = time.perf_counter()
    st = ra.MappedStream(qi, f1, f2, device=0, batch_units=1 << 18, threads=thr, names=names)
    nh = 0
    for b in st:
        nh += b.n_hits
    dt = time.perf_counter() - t
    ss = st.stats(); st.close()
    print(json.dumps({"what": "stream", "files": tag, "threads": thr, "names": names, "M_pairs_s": round(n / dt / 1e6, 2), "s": round(dt, 4),
                      **{k: round(v, 4) for k, v in ss.items() if k in ("read_s", "first_batch_s", "parse_cpu_s", "copy_cpu_s", "caller_wait_s")}}), flush=True)

for tag, il in (("default", False), ("interleaved", True)):
    f1, f2 = write(tag, il)
    reader(f1, f2, 32, tag)                       # (first pass over the fresh files: not comparable, printed for the record)
    for thr in (16, 32, 48, 64, 96, 128):
        reader(f1, f2, thr, tag)
    for thr in (32, 64, 96):
        stream(f1, f2, thr, tag)
    stream(f1, f2, 64, tag, names=True)
    os.remove(f1); os.remove(f2)
