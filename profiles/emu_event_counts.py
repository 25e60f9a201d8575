#!/usr/bin/env python3
"""Dynamic event counts per read of the stage-A mapper, from the lane-emulation build (-DQM_PROFILE).
Run on the CPU: python profiles/emu_event_counts.py [genes] [pairs].  The workload is a scaled-down
bench workload (same generator, 1% substitutions, 2x100 bp)."""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
import rapmap_amd as ra  # noqa: E402
from rapmap_amd import synth  # noqa: E402
from oracle import q5  # noqa: E402
import emu  # noqa: E402

NAMES = ["find_kmer calls", "slot loads", "setup_strand", "probe_window calls", "probe_window positions",
         "extend_wide calls", "extend_wide suffix lanes", "extend_wide 8B steps", "extend literal", "extend_wide width==1",
         "rank_sort calls", "rank_sort n", "single_interval calls", "single_interval n", "multi_interval calls",
         "multi_interval m", "cmp_from steps", "get_sa_hits calls", "extension calls"]


def main():
    genes = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    td = tempfile.mkdtemp(prefix="qmprof")
    lib = os.path.join(td, "libqm_emu_prof.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DQM_PROFILE", "-Wno-unused", "-o", lib,
                           os.path.join(ROOT, "tests", "emu", "qm_emu.cpp")])
    emu._LIB = lib
    emu._SRC = [lib]
    names, txps = synth.make_transcriptome(genes, seed=42)
    fa = os.path.join(td, "t.fa"); synth.write_fasta(fa, names, txps)
    ra.build_index(fa, os.path.join(td, "idx"), threads=8)
    ix = q5.load(os.path.join(td, "idx"))
    s1, s2, off, _ = synth.make_reads(txps, pairs, seed=43, read_len=100, err=0.01)
    em = emu.Emu(ix)
    em.lib.qe_prof.restype = C.POINTER(C.c_uint64)
    er = em.map(s1, off, s2, off)
    pr = em.lib.qe_prof()
    nreads = 2 * pairs
    print("genes %d, pairs %d, hits/pair %.3f" % (genes, pairs, er.hits.size / pairs))
    for i, nm in enumerate(NAMES):
        print("%-28s %10d   per read %8.3f" % (nm, pr[i], pr[i] / nreads))


if __name__ == "__main__":
    main()
