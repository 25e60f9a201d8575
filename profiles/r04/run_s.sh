#!/bin/bash
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python - <<'PY' > $OUT/stream.log 2>&1
import json, os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
import bench, rapmap_amd as ra
from rapmap_amd import synth
n = 10_000_000; L = 100; reps = 4
dev = torch.device("cuda", 0)
idx = bench.build_or_reuse_index(40000, 42, 31, 0, 1, "/dev/shm")
qi = ra.QuasiIndex(idx)
text, starts, lens = bench.load_text_to_gpu(qi, dev)
s1, s2, off = bench.make_reads_gpu(text, starts, lens, n, 43, dev, read_len=L)
h1 = s1[: n * L].cpu().numpy(); h2 = s2[: n * L].cpu().numpy()
del s1, s2, text; torch.cuda.empty_cache()
keep = ra.QuasiMapper(qi, 0)
ra.reserve_stream_memory(2048 << 20)
d = "/tmp/qmap_e2e_p"; os.makedirs(d, exist_ok=True)
f1, f2 = d + "/r1.fq", d + "/r2.fq"
for r in range(reps):
    synth.write_fastq(f1, h1, n, L, 1, start=r * n, append=r > 0); synth.write_fastq(f2, h2, n, L, 2, start=r * n, append=r > 0)
def stream(thr, names, env=None):
    for k, v in (env or {}).items(): os.environ[k] = v
    t = time.perf_counter()
    st = ra.MappedStream(qi, f1, f2, device=0, batch_units=1 << 18, threads=thr, names=names)
    nh = 0
    for b in st: nh += b.n_hits
    dt = time.perf_counter() - t
    ss = st.stats(); st.close()
    for k in (env or {}): os.environ.pop(k)
    print(json.dumps({"threads": thr, "names": names, "env": env, "M_pairs_s": round(n * reps / dt / 1e6, 2), "s": round(dt, 4), "hits": int(nh),
                      **{k: round(v, 4) for k, v in ss.items() if k in ("read_s", "first_batch_s", "parse_cpu_s", "copy_cpu_s", "caller_wait_s", "map_s")}}), flush=True)
for rnd in range(4):
    for c in ("3", "4"):
        stream(24, False, {"QM_STREAM_CTX_PER_DEVICE": c}); stream(24, False, {"QM_STREAM_CTX_PER_DEVICE": c, "QM_STREAM_NO_PACK": "1"})
for rnd in range(2):
    stream(16, False, {"QM_STREAM_CTX_PER_DEVICE": "3"}); stream(16, False, {"QM_STREAM_CTX_PER_DEVICE": "3", "QM_STREAM_NO_PACK": "1"})
    stream(24, True, {"QM_STREAM_CTX_PER_DEVICE": "3"}); stream(24, True, {"QM_STREAM_CTX_PER_DEVICE": "3", "QM_STREAM_NO_PACK": "1"})
os.remove(f1); os.remove(f2)
PY
grep -v amdgpu $OUT/stream.log | tail -26
