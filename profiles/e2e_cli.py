#!/usr/bin/env python3
"""End-to-end rates of the CLI (FASTQ on tmpfs -> hits / SAM), SURVEY.md section 8d "end-to-end M pairs/s".
Run on the GPU box: python profiles/e2e_cli.py [genes] [pairs] [threads].  Synthetic 2x100 bp FASTQ files are written
with fixed-width records by numpy; the index is the bench's generator at `genes` genes."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rapmap_amd as ra  # noqa: E402
from rapmap_amd import synth  # noqa: E402


write_fastq = synth.write_fastq


def main():
    genes = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
    threads = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    chunks = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1 << 20]
    thr_list = [int(x) for x in sys.argv[5].split(",")] if len(sys.argv) > 5 else [threads]
    d = tempfile.mkdtemp(prefix="qme2e", dir="/dev/shm")
    names, txps = synth.make_transcriptome(genes, seed=42)
    fa = os.path.join(d, "t.fa"); synth.write_fasta(fa, names, txps)
    t = time.time(); ra.build_index(fa, os.path.join(d, "idx"), threads=threads)
    print("[e2e] index: %d transcripts, built in %.1fs" % (len(txps), time.time() - t), flush=True)
    s1, s2, off, _ = synth.make_reads(txps, pairs, seed=43)
    f1 = os.path.join(d, "r1.fq"); f2 = os.path.join(d, "r2.fq")
    write_fastq(f1, s1, pairs, 100, 1); write_fastq(f2, s2, pairs, 100, 2)
    print("[e2e] FASTQ: 2 x %.0f MB on tmpfs" % (os.path.getsize(f1) / 1e6), flush=True)
    env = dict(os.environ, PYTHONPATH=ROOT)
    runs = []
    for ch in chunks:
        for th in thr_list:
            runs.append(("FASTQ -> hits (-n) chunk %d t%d" % (ch, th), ["-n"], ch, th))
    runs.append(("FASTQ -> SAM on tmpfs (-o)", ["-o", os.path.join(d, "out.sam")], chunks[-1], thr_list[-1]))
    runs.append(("FASTQ -> SAM.gz on tmpfs (-o -x)", ["-o", os.path.join(d, "out.sam.gz"), "-x"], chunks[-1], thr_list[-1]))
    for label, extra, ch, th in runs:
        base = [sys.executable, "-m", "rapmap_amd", "quasimap", "-i", os.path.join(d, "idx"), "-1", f1, "-2", f2, "-t", str(th), "--chunk", str(ch)]
        best = None
        for rep in range(2):
            t = time.time()
            r = subprocess.run(base + extra, env=env, cwd=ROOT, capture_output=True, text=True)
            dt = time.time() - t
            if r.returncode != 0:
                print(r.stderr[-2000:]); sys.exit(1)
            best = dt if best is None else min(best, dt)
        tail = [l for l in r.stderr.splitlines() if "Elapsed" in l or "Final" in l or l.startswith("stream:")]
        print("[e2e] %-28s %6.2f s  -> %6.2f M pairs/s end to end (process start to exit; %s)" % (label, best, pairs / best / 1e6, "; ".join(tail)), flush=True)
    for f in ("out.sam", "out.sam.gz"):
        if os.path.exists(os.path.join(d, f)):
            print("[e2e] %s size %.0f MB" % (f, os.path.getsize(os.path.join(d, f)) / 1e6))
    subprocess.run(["rm", "-rf", d])


if __name__ == "__main__":
    main()
