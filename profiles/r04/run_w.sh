#!/bin/bash
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python - <<'PY' > $OUT/probe.txt 2>&1
import json, os, subprocess, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
import bench, rapmap_amd as ra
n = 8_000_000
idx = bench.build_or_reuse_index(40000, 42, 31, 0, 1, "/dev/shm")
qi = ra.QuasiIndex(idx); dev = torch.device("cuda", 0)
text, starts, lens = bench.load_text_to_gpu(qi, dev)
s1, s2, off = bench.make_reads_gpu(text, starts, lens, n, 43, dev)
h1 = s1[: n * 100].cpu().numpy(); h2 = s2[: n * 100].cpu().numpy()
os.makedirs("/tmp/cp", exist_ok=True)
exe = bench.build_compat_bench("/tmp/cp")
with open("/tmp/cp/reads.bin", "wb") as f:
    f.write(h1.tobytes()); f.write(h2.tobytes())
del s1, s2, text
torch.cuda.empty_cache()
def run(T, env=None):
    e = dict(os.environ); e.update(env or {})
    p = subprocess.run([exe, idx, "/tmp/cp/reads.bin", str(n), "100", str(T), "10000", "--repeat", "4"], capture_output=True, text=True, env=e, timeout=300)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    j = json.loads(line[-1]) if line else {}
    print("threads %2d %s: %.2f M pairs/s (join %.3f s)" % (T, env or "", j.get("mpairs_per_s", -1), j.get("thread_join_seconds", -1)), flush=True)
for T in (32, 16, 64):
    for env in (None, {"QMAP_COMPAT_CONTEXTS": "3"}, {"QMAP_COMPAT_CONTEXTS": "4"}, {"QMAP_COMPAT_LINGER_US": "600"}, {"QMAP_COMPAT_CONTEXTS": "3", "QMAP_COMPAT_LINGER_US": "600"}, {"QM_HOST_CHUNK": "32768"}, {"QM_HOST_CHUNK": "32768", "QMAP_COMPAT_CONTEXTS": "3"}):
        run(T, env)
PY
grep -v amdgpu.ids $OUT/probe.txt
