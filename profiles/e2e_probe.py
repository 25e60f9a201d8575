"""Where the CLI loop's time goes (GPU box): reader vs map+fetch per batch, 8 M pairs of FASTQ on tmpfs.  python profiles/e2e_probe.py"""
import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "profiles"))
import e2e_cli
import rapmap_amd as ra
from rapmap_amd import synth
n = 8_000_000; L = 100
d = "/dev/shm/qmprobe"; os.makedirs(d, exist_ok=True)
names, txps = synth.make_transcriptome(4000, seed=42)
fa = d + "/t.fa"; synth.write_fasta(fa, names, txps)
ra.build_index(fa, d + "/idx", threads=32)
s1, s2, off, _ = synth.make_reads(txps, n, seed=43, read_len=L, err=0.01) if n <= 2_000_000 else (None, None, None, None)
if s1 is None:
    a, b, o, _ = synth.make_reads(txps, 2_000_000, seed=43, read_len=L, err=0.01)
    s1 = np.tile(a, 4); s2 = np.tile(b, 4)
e2e_cli.write_fastq(d + "/r1.fq", s1, n, L, 1); e2e_cli.write_fastq(d + "/r2.fq", s2, n, L, 2)
qi = ra.QuasiIndex(d + "/idx"); mp = ra.QuasiMapper(qi, 0)
for thr in (16, 64, 128):
    for rep in range(2):
        t0 = time.time(); rd = ra.FastxReader(d + "/r1.fq", d + "/r2.fq", threads=thr); tot = 0; tr = 0; tm = 0
        ta = time.time()
        for b in rd.chunks(1 << 20):
            tb = time.time(); tr += tb - ta
            r = mp.map_pairs(b.seq1, b.off1, b.seq2, b.off2)
            ta = time.time(); tm += ta - tb
            tot += b.n
        rd.close()
        print("threads %d: total %.3f s, reader %.3f s, map+fetch %.3f s (pairs %d)" % (thr, time.time() - t0, tr, tm, tot), flush=True)
import shutil; shutil.rmtree(d)
