"""FASTQ (tmpfs) -> hits through qm_stream_* on the bench workload (configs[1] index, 2x100 bp pairs): sweep of ingest worker
counts, batch sizes, names kept or not; prints the stream's per-phase seconds.  python profiles/r03_e2e_stream.py [pairs] [genes]"""
import json, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import rapmap_amd as ra
from rapmap_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
genes = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
L = 100
dev = torch.device("cuda", 0)
idx_dir = bench.build_or_reuse_index(genes, 42, 31, 0, 1, "/dev/shm")
qi = ra.QuasiIndex(idx_dir)
text, starts, lens = bench.load_text_to_gpu(qi, dev)
s1, s2, off = bench.make_reads_gpu(text, starts, lens, n, 43, dev, read_len=L)
d = os.environ.get("E2E_DIR", "/tmp/qmap_e2e"); os.makedirs(d, exist_ok=True)
f1, f2 = d + "/r1.fq", d + "/r2.fq"
synth.write_fastq(f1, s1[: n * L].cpu().numpy(), n, L, 1); synth.write_fastq(f2, s2[: n * L].cpu().numpy(), n, L, 2)
del s1, s2, text
torch.cuda.empty_cache()
print("files: 2 x %.0f MB, host threads %d" % (os.path.getsize(f1) / 1e6, os.cpu_count()), flush=True)
if os.environ.get("E2E_RESERVE", "1") != "0":
    ra.reserve_stream_memory(768 << 20)
keep = ra.QuasiMapper(qi, 0)      # keeps the index replica resident between the runs (as the CLI does)

def run(threads, batch, names, label=""):
    t = time.perf_counter()
    st = ra.MappedStream(qi, f1, f2, device=0, batch_units=batch, threads=threads, names=names)
    nh = 0; nb = 0
    for b in st:
        nh += b.n_hits; nb += 1
    dt = time.perf_counter() - t
    ss = st.stats(); st.close()
    print(json.dumps({"threads": threads, "batch": batch, "names": names, "M_pairs_s": round(n / dt / 1e6, 2), "s": round(dt, 4), "batches": nb, "hits": nh,
                      **{k: round(v, 4) for k, v in ss.items()}, "label": label}), flush=True)

run(32, 1 << 18, True, "first pass over freshly written files")
for thr in (16, 24, 32, 40, 48, 64):
    run(thr, 1 << 18, True)
for batch in (1 << 16, 1 << 17, 1 << 19):
    run(32, batch, True)
run(32, 1 << 18, False)
run(32, 1 << 17, False)
for cpd in ("3", "4"):
    os.environ["QM_STREAM_CTX_PER_DEVICE"] = cpd
    run(32, 1 << 18, True, cpd + " contexts")
    run(32, 1 << 17, True, cpd + " contexts")
    run(48, 1 << 17, True, cpd + " contexts")
del os.environ["QM_STREAM_CTX_PER_DEVICE"]
os.environ["QM_INGEST_PREAD"] = "1"
run(32, 1 << 18, True, "pread")
run(48, 1 << 18, True, "pread")
del os.environ["QM_INGEST_PREAD"]
# reader alone (malloc'd slots, no GPU work): what the ingest engine delivers
for thr in (16, 32, 48, 64):
    t = time.perf_counter(); rd = ra.FastxReader(f1, f2, threads=thr); tot = 0
    for b in rd.chunks(1 << 18):
        tot += b.n
    rd.close(); dt = time.perf_counter() - t
    print(json.dumps({"reader_only_threads": thr, "M_pairs_s": round(tot / dt / 1e6, 2), "GB_s": round(2 * os.path.getsize(f1) / dt / 1e9, 2)}), flush=True)
for f in (f1, f2):
    os.remove(f)
