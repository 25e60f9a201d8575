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
def run(T, chunk, extra=(), env=None, use=n):
    e = dict(os.environ); e.update(env or {}); e["COMPAT_BENCH_VERBOSE"] = "1"
    p = subprocess.run([exe, idx, "/tmp/cp/reads.bin", str(n), "100", str(T), str(chunk), "--use", str(use)] + list(extra), capture_output=True, text=True, env=e, timeout=300)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    j = json.loads(line[-1]) if line else {}
    print("threads %2d chunk %6d %s %s: %.2f M pairs/s" % (T, chunk, " ".join(extra), env or "", j.get("mpairs_per_s", -1)), flush=True)
    print("\n".join(l for l in p.stderr.splitlines() if l.startswith("[compat_bench]")), flush=True)
run(32, 10000, ["--repeat", "3"])
run(8, 10000, ["--repeat", "2"])
PY
grep -v amdgpu.ids $OUT/probe.txt
