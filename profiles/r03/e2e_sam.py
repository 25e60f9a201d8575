"""FASTQ (local filesystem) -> SAM file through qm_stream_* + qm_sam_writer_* on the bench workload: what `quasimap -o` does, without
the process start.  python profiles/r03/e2e_sam.py [pairs]"""
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
d = os.environ.get("E2E_DIR", "/tmp/qmap_e2e"); os.makedirs(d, exist_ok=True)
f1, f2 = d + "/r1.fq", d + "/r2.fq"
synth.write_fastq(f1, s1[: n * L].cpu().numpy(), n, L, 1); synth.write_fastq(f2, s2[: n * L].cpu().numpy(), n, L, 2)
del s1, s2, text
keep = ra.QuasiMapper(qi, 0)
for fthr, gz in ((8, False), (16, False), (32, False), (64, False), (32, True)):
    out = d + ("/out.sam.gz" if gz else "/out.sam")
    fd = os.open(out, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    t = time.perf_counter()
    w = ra.SamWriter(qi, fd, max_num_hits=200, threads=fthr, gzip=gz)
    w.header()
    st = ra.MappedStream(qi, f1, f2, device=0, batch_units=1 << 18, threads=32)
    tput = 0.0
    for b in st:
        t1 = time.perf_counter(); w.put(b, b.hit_offsets, b.hits); tput += time.perf_counter() - t1
    ss = st.stats(); st.close()
    nb = w.close(); os.close(fd)
    dt = time.perf_counter() - t
    print(json.dumps({"formatter_threads": fthr, "gzip": gz, "M_pairs_s": round(n / dt / 1e6, 2), "s": round(dt, 3), "sam_GB": round(nb / 1e9, 2),
                      "GB_s": round(nb / dt / 1e9, 2), "put_s": round(tput, 3), "read_s": round(ss["read_s"], 3), "caller_wait_s": round(ss["caller_wait_s"], 3)}), flush=True)
    os.remove(out)
os.remove(f1); os.remove(f2)
