"""compat face probe (GPU box): config-2 index, N pairs of the bench's generator, compat_bench at several thread counts with its
prefetch / loop split.   python profiles/r05/compat_probe.py [pairs] [threads,threads,...] [extra compat_bench flags]"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, rapmap_amd as ra
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
threads = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,8,32").split(",")]
extra = sys.argv[3:]
idx = bench.build_or_reuse_index(40000, 42, 31, 0, 1, "/dev/shm")
qi = ra.QuasiIndex(idx); dev = torch.device("cuda", 0)
text, starts, lens = bench.load_text_to_gpu(qi, dev)
s1, s2, off = bench.make_reads_gpu(text, starts, lens, n, 43, dev)
mp = ra.QuasiMapper(qi, 0)
r = mp.map_device(n, s1.data_ptr(), off.data_ptr(), s2.data_ptr(), off.data_ptr(), 100, fetch=True)
want = bench.compat_digest(r.hit_offsets, r.hits)
h1 = s1[: n * 100].cpu().numpy(); h2 = s2[: n * 100].cpu().numpy()
os.makedirs("/tmp/cp", exist_ok=True)
open("/tmp/cp/idx.txt", "w").write(idx)
exe = bench.build_compat_bench("/tmp/cp")
with open("/tmp/cp/reads.bin", "wb") as f:
    f.write(h1.tobytes()); f.write(h2.tobytes())
mp.close()
for T in threads:
    p = subprocess.run([exe, idx, "/tmp/cp/reads.bin", str(n), "100", str(T), "10000"] + extra, capture_output=True, text=True, timeout=240)
    if os.environ.get("QMAP_COMPAT_DEBUG") or os.environ.get("COMPAT_BENCH_VERBOSE"):
        print("\n".join(p.stderr.splitlines()[-int(os.environ.get("PROBE_TAIL", "24")):]))
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(T, "FAILED", p.stdout[-300:], p.stderr[-300:]); continue
    j = json.loads(line[-1])
    print("threads %2d: %.2f M pairs/s  prefetch %.3f thread-s  loop %.3f thread-s  (%.1f / %.1f us per pair)  parity %s" % (
        T, j["mpairs_per_s"], j["prefetch_thread_s"], j["loop_thread_s"], j["prefetch_thread_s"] / n * 1e6, j["loop_thread_s"] / n * 1e6, j["digest"] == want), flush=True)
