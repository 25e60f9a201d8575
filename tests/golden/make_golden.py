#!/usr/bin/env python3
"""Generates tests/golden/synth_small/: a small synthetic transcriptome, 2x100 bp read pairs plus
adversarial reads, and -- when a reference binary is available -- the SAM the reference writes.

Provenance of expected_*.sam.gz: produced with the probe build of the *unmodified* reference sources
that the survey stage left in this container (/tmp/oracle/build/rapmap; it
was linked against a stand-in for the un-vendored cereal headers, which only touches (de)serialisation).
This round's rules do not allow making such a build, so it is neither rebuilt nor shipped; its
outputs are kept as corroborating vectors next to the SURVEY-recorded sample_data digest.
Run: python tests/golden/make_golden.py [path/to/rapmap]
"""
import gzip
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from rapmap_amd import synth  # noqa: E402

OUT = os.path.join(HERE, "synth_small")
B = np.frombuffer(b"ACGT", dtype=np.uint8)


def rc(b):
    return bytes(b.translate(bytes.maketrans(b"ACGTNacgtn", b"TGCANtgcan"))[::-1])


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/tmp/oracle/build/rapmap"
    rng = np.random.default_rng(7)
    names, txps = synth.make_transcriptome(90, seed=11)
    # repeat families: one 160-mer shared by 260 transcripts (> maxNumHits 200), one by 1100 (> maxInterval 1000)
    for fam, copies in (("REPA", 260), ("REPB", 1100)):
        core = B[rng.integers(0, 4, 160)]
        for i in range(copies):
            l, r = B[rng.integers(0, 4, int(rng.integers(20, 60)))], B[rng.integers(0, 4, int(rng.integers(20, 60)))]
            txps.append(np.concatenate([l, core, r]))
            names.append("%s.%d" % (fam, i))
    # poly-A tails, lower case, an N run and a duplicate to exercise the indexer's text rules
    txps.append(np.concatenate([B[rng.integers(0, 4, 300)], np.full(25, ord("A"), np.uint8)])); names.append("POLYA.1")
    dup = txps[3].copy(); txps.append(dup); names.append("DUP.of.3")
    low = np.frombuffer(txps[5].tobytes().lower(), dtype=np.uint8).copy(); txps.append(low); names.append("LOWER.5")
    withn = txps[7].copy(); withn[100:104] = ord("N"); txps.append(withn); names.append("WITHN.7 extra header text")
    os.makedirs(OUT, exist_ok=True)
    fa = os.path.join(OUT, "txome.fa")
    synth.write_fasta(fa, names, txps)
    with open(fa, "rb") as fi, gzip.open(fa + ".gz", "wb", compresslevel=9) as fo:
        fo.write(fi.read())

    base = txps[:400]
    s1, s2, off, truth = synth.make_reads(base, 3000, seed=13, read_len=100, err=0.01)
    r1 = [s1[off[i]:off[i + 1]].tobytes() for i in range(3000)]
    r2 = [s2[off[i]:off[i + 1]].tobytes() for i in range(3000)]
    # error-free and high-error pairs
    for e, cnt, sd in ((0.0, 300, 14), (0.05, 300, 15)):
        a1, a2, o, _ = synth.make_reads(base, cnt, seed=sd, read_len=100, err=e)
        r1 += [a1[o[i]:o[i + 1]].tobytes() for i in range(cnt)]
        r2 += [a2[o[i]:o[i + 1]].tobytes() for i in range(cnt)]
    # N's sprinkled in
    a1, a2, o, _ = synth.make_reads(base, 300, seed=16, read_len=100, err=0.01, n_rate=0.01)
    r1 += [a1[o[i]:o[i + 1]].tobytes() for i in range(300)]
    r2 += [a2[o[i]:o[i + 1]].tobytes() for i in range(300)]
    # variable read lengths (60..128), incl. shorter than k
    for i in range(200):
        t = base[int(rng.integers(0, len(base)))]
        la, lb = int(rng.integers(25, 129)), int(rng.integers(25, 129))
        fl = min(t.size, max(la, lb) + int(rng.integers(0, 150)))
        st = int(rng.integers(0, t.size - fl + 1))
        frag = t[st:st + fl].tobytes()
        r1.append(frag[:la]); r2.append(rc(frag[-lb:]))
    # repeat-family reads (tooManyHits / maxInterval paths), both orientations
    for fam in (txps[-(4 + 1100 + 260)], txps[-(4 + 1100)]):
        pass
    repA = [t for n, t in zip(names, txps) if n.startswith("REPA")]
    repB = [t for n, t in zip(names, txps) if n.startswith("REPB")]
    for fam in (repA, repB):
        for i in range(60):
            t = fam[i].tobytes()
            lo = 0 if i % 3 else 10
            a = t[lo:lo + 100]; b = rc(t[-100:])
            if i % 2:
                a, b = b, a
            r1.append(a); r2.append(b)
    # adversarial singles
    adv = [
        (b"A" * 100, b"T" * 100),                                  # homopolymers
        (b"ACGT" * 25, b"N" * 100),                                # all-N mate
        (base[0][:100].tobytes().lower(), rc(base[0][150:250].tobytes())),           # lower case
        (base[1][:50].tobytes() + b"N" + base[1][51:100].tobytes(), rc(base[1][120:220].tobytes())),
        (base[2][:31].tobytes() + b"N" + base[2][32:100].tobytes(), rc(base[2][120:220].tobytes())),   # N right after first k-mer
        (base[3][:100].tobytes().replace(b"A", b"R", 1), rc(base[3][120:220].tobytes())),            # IUPAC R
        (base[4][:100].tobytes().replace(b"T", b"U"), rc(base[4][120:220].tobytes())),               # RNA U's
        (base[5][:30].tobytes(), rc(base[5][100:130].tobytes())),                                    # shorter than k
        (base[6][:31].tobytes(), rc(base[6][100:131].tobytes())),                                    # exactly k
        (base[7][-60:].tobytes() + base[8][:40].tobytes(), rc(base[8][50:150].tobytes())),           # spans the '$' between txps
        (rc(base[9][:100].tobytes()), base[9][150:250].tobytes()),                                   # dovetail-ish / swapped
        (base[10][:100].tobytes(), base[10][20:120].tobytes()),                                      # same strand
        (base[11][:100].tobytes(), rc(base[12][:100].tobytes())),                                    # mates on different genes
        (b"", base[13][:100].tobytes()),                                                             # empty mate
    ]
    for a, b in adv:
        if len(a) == 0:
            a = b"N"      # FASTQ parsers reject empty records; keep a 1-base stand-in
        r1.append(a); r2.append(b)
    n = len(r1)
    with gzip.open(os.path.join(OUT, "reads_1.fastq.gz"), "wb", compresslevel=9) as f1, \
            gzip.open(os.path.join(OUT, "reads_2.fastq.gz"), "wb", compresslevel=9) as f2:
        for i in range(n):
            f1.write(b"@p%d/1\n%s\n+\n%s\n" % (i, r1[i], b"I" * len(r1[i])))
            f2.write(b"@p%d/2\n%s\n+\n%s\n" % (i, r2[i], b"I" * len(r2[i])))
    print("transcripts", len(txps), "pairs", n)

    if not os.path.exists(ref):
        os.remove(fa)
        print("no reference binary at", ref, "- inputs written, expected outputs left untouched")
        return
    with tempfile.TemporaryDirectory() as td:
        for nm in ("reads_1", "reads_2"):
            with gzip.open(os.path.join(OUT, nm + ".fastq.gz"), "rb") as g, open(os.path.join(td, nm + ".fq"), "wb") as o:
                o.write(g.read())
        idx = os.path.join(td, "idx")
        # The reference indexer occasionally hands the FASTA chunks to its consumer out of order
        # (observed here: the last chunk first), which permutes transcript ids.  Only an index whose
        # transcript order equals the file order is a usable fixture, so rebuild until it is.
        import hashlib
        import shutil
        sys.path.insert(0, os.path.join(HERE, "..", ".."))
        from oracle import q5
        for attempt in range(10):
            shutil.rmtree(idx, ignore_errors=True)
            subprocess.check_call([ref, "quasiindex", "-t", fa, "-i", idx], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            got = q5.read_txpinfo(idx)[0]
            kept = set(got)
            if got == [n.split(" ")[0] for n in names if n.split(" ")[0] in kept] and "DUP.of.3" not in kept:
                break
            print("reference indexer permuted the transcripts (attempt %d), retrying" % attempt)
        else:
            raise SystemExit("no index in file order after 10 attempts")
        # byte-level index files of the reference for the indexer test
        with open(os.path.join(OUT, "expected_index.md5"), "w") as f:
            for fn in ("sa.bin", "txpInfo.bin", "rsd.bin"):
                f.write("%s  %s\n" % (hashlib.md5(open(os.path.join(idx, fn), "rb").read()).hexdigest(), fn))
        variants = {
            "default": [],
            "noStrictCheck": ["--noStrictCheck"],
            "z0.9": ["-z", "0.9"],
            "m3": ["-m", "3"],
            "noOrphans": ["--noOrphans"],
            "noSensitive": ["--noSensitive"],
            "fuzzy": ["-f"],
            "fuzzy_noOrphans_m3": ["-f", "--noOrphans", "-m", "3"],
        }
        os.remove(fa)     # only the .gz is committed
        for name, flags in variants.items():
            sam = os.path.join(td, name + ".sam")
            subprocess.check_call([ref, "quasimap", "-q", "-t", "1", "-i", idx, "-1", os.path.join(td, "reads_1.fq"),
                                   "-2", os.path.join(td, "reads_2.fq"), "-o", sam] + flags,
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            # keep every column except SEQ (col 10): the sequences are in reads_*.fastq.gz already
            body = b"".join(b"\t".join(l.split(b"\t")[:9] + l.split(b"\t")[10:])
                            for l in open(sam, "rb") if not l.startswith(b"@"))
            with gzip.open(os.path.join(OUT, "expected_%s.noseq.sam.gz" % name), "wb", compresslevel=9) as g:
                g.write(body)
            print(name, "records", body.count(b"\n"))


if __name__ == "__main__":
    main()
