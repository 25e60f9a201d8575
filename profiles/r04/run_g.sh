#!/bin/bash
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "beyond_512 or edge" > $OUT/pytest_long.log 2>&1; tail -3 $OUT/pytest_long.log
python - <<'PY' > $OUT/probe.txt 2>&1
import json, os, subprocess, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
import bench, rapmap_amd as ra
n = 6_000_000
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
    e = dict(os.environ); e.update(env or {})
    p = subprocess.run([exe, idx, "/tmp/cp/reads.bin", str(n), "100", str(T), str(chunk), "--use", str(use)] + list(extra), capture_output=True, text=True, env=e, timeout=300)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    j = json.loads(line[-1]) if line else {}
    print("threads %2d chunk %6d %s %s: %.2f M pairs/s  prefetch %.3f loop %.3f thread-s" % (T, chunk, " ".join(extra), env or "", j.get("mpairs_per_s", -1), j.get("prefetch_thread_s", -1), j.get("loop_thread_s", -1)), flush=True)
    return p.stderr
for ch in (10000, 40000, 160000, 640000):
    run(1, ch, use=1_920_000)
err = run(32, 10000, ["--repeat", "3"], {"QMAP_COMPAT_DEBUG": "1"})
lines = [l for l in err.splitlines() if l.startswith("[qmap service]")]
print("\n".join(lines[:12])); print("..."); print("\n".join(lines[-40:]))
run(32, 10000, ["--repeat", "3"])
run(64, 10000, ["--repeat", "3"])
run(16, 10000, ["--repeat", "3"])
PY
tail -70 $OUT/probe.txt
