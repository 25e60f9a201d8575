"""First-pass vs second-pass cost of the pipelined stream in one process (plain / after torch init / with a large host array
held): python profiles/e2e_probe2.py plain|torch|torchbig.  8 M synthetic pairs on tmpfs, two passes, stream stats printed."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
mode = sys.argv[1]
if mode != "plain":
    import torch
    torch.cuda.init()
    if mode == "torchbig":
        x = torch.empty(10_000_000 * 200, dtype=torch.uint8, device="cuda"); y = x.cpu().numpy()   # a big host array like bench holds
import numpy as np
import rapmap_amd as ra
from rapmap_amd import synth
d = "/dev/shm/e2et"; os.makedirs(d, exist_ok=True)
idx = os.path.join(d, "idx")
if not os.path.exists(os.path.join(idx, "sa.bin")) and not os.path.exists(os.path.join(d, "DONE")):
    names, txps = synth.make_transcriptome(4000, seed=42)
    fa = os.path.join(d, "t.fa"); synth.write_fasta(fa, names, txps)
    ra.build_index(fa, idx, threads=32)
    s1, s2, off, _ = synth.make_reads(txps, 8_000_000, seed=43)
    synth.write_fastq(os.path.join(d, "r1.fq"), s1, 8_000_000, 100, 1); synth.write_fastq(os.path.join(d, "r2.fq"), s2, 8_000_000, 100, 2)
    open(os.path.join(d, "DONE"), "w").write("ok")
qi = ra.QuasiIndex(idx)
for rep in range(2):
    t = time.perf_counter()
    st = ra.MappedStream(qi, os.path.join(d, "r1.fq"), os.path.join(d, "r2.fq"), batch_units=1 << 18, threads=64)
    nh = 0
    for b in st: nh += b.n_hits
    dt = time.perf_counter() - t
    print(mode, "rep", rep, "%.3f s" % dt, "%.1f M pairs/s" % (8 / dt), st.stats(), flush=True)
    st.close()
