#!/usr/bin/env python3
"""More reference outputs on the committed synth_small inputs, generated while the survey stage's probe build of the
unmodified reference (/tmp/oracle/build/rapmap, see make_golden.py for its provenance) is still in this container:
  * option sets of SURVEY.md section 8 rows that are not built yet (-s selective alignment and friends, -c), so
    that the next rounds have vectors to pin their oracle on;
  * option sets that are built but had no reference vector (--noDovetail on paired hits, single-end input).
Also writes a second, small read set with insertions/deletions (reads_indel_*.fastq.gz): substitutions alone never
exercise the gapped part of the ksw2 extension alignment.
Existing fixtures are not touched.  Run: python tests/golden/make_golden_next.py [path/to/rapmap]"""
import gzip
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import q5  # noqa: E402

SRC = os.path.join(HERE, "synth_small")
OUT = os.path.join(SRC, "next")
B = b"ACGT"


def rc(b):
    return bytes(b.translate(bytes.maketrans(b"ACGTNacgtn", b"TGCANtgcan"))[::-1])


def read_fasta(path):
    names, seqs = [], []
    with gzip.open(path, "rb") as f:
        for l in f:
            l = l.rstrip()
            if l.startswith(b">"):
                names.append(l[1:].decode()); seqs.append([])
            else:
                seqs[-1].append(l)
    return names, [b"".join(s) for s in seqs]


def mutate_indel(rng, s):
    """1-2 small insertions / deletions plus ~1% substitutions"""
    s = bytearray(s)
    for _ in range(int(rng.integers(1, 3))):
        p = int(rng.integers(20, max(21, len(s) - 20)))
        ln = int(rng.integers(1, 4))
        if rng.random() < 0.5:
            del s[p:p + ln]
        else:
            s[p:p] = bytes(B[int(x)] for x in rng.integers(0, 4, ln))
    for i in range(len(s)):
        if rng.random() < 0.01:
            s[i] = B[(B.index(bytes([s[i]]).upper()) + int(rng.integers(1, 4))) % 4] if bytes([s[i]]).upper() in (b"A", b"C", b"G", b"T") else s[i]
    return bytes(s)


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/tmp/oracle/build/rapmap"
    if not os.path.exists(ref):
        sys.exit("no reference binary at %s" % ref)
    os.makedirs(OUT, exist_ok=True)
    names, txps = read_fasta(os.path.join(SRC, "txome.fa.gz"))
    rng = np.random.default_rng(21)
    base = [t.upper() for t in txps[:400] if len(t) >= 400 and b"N" not in t.upper()]
    r1, r2 = [], []
    for i in range(1500):
        t = base[int(rng.integers(0, len(base)))]
        fl = int(rng.integers(220, 320))
        st = int(rng.integers(0, len(t) - fl + 1))
        frag = t[st:st + fl]
        a, b = mutate_indel(rng, frag[:104])[:100], mutate_indel(rng, rc(frag[-104:]))[:100]
        if rng.random() < 0.5:
            a, b = b, a
        r1.append(a); r2.append(b)
    for nm, rr in (("reads_indel_1", r1), ("reads_indel_2", r2)):
        with gzip.open(os.path.join(OUT, nm + ".fastq.gz"), "wb", compresslevel=9) as f:
            for i, s in enumerate(rr):
                f.write(b"@q%d/%s\n%s\n+\n%s\n" % (i, nm[-1:].encode(), s, b"I" * len(s)))
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, "txome.fa")
        with gzip.open(os.path.join(SRC, "txome.fa.gz"), "rb") as g, open(fa, "wb") as o:
            o.write(g.read())
        for nm, src in (("reads_1", SRC), ("reads_2", SRC), ("reads_indel_1", OUT), ("reads_indel_2", OUT)):
            with gzip.open(os.path.join(src, nm + ".fastq.gz"), "rb") as g, open(os.path.join(td, nm + ".fq"), "wb") as o:
                o.write(g.read())
        idx = os.path.join(td, "idx")
        want = open(os.path.join(SRC, "expected_index.md5")).read()
        import hashlib
        for attempt in range(20):      # the reference indexer sometimes permutes the transcripts (see make_golden.py)
            shutil.rmtree(idx, ignore_errors=True)
            subprocess.check_call([ref, "quasiindex", "-t", fa, "-i", idx], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            got = "".join("%s  %s\n" % (hashlib.md5(open(os.path.join(idx, fn), "rb").read()).hexdigest(), fn) for fn in ("sa.bin", "txpInfo.bin", "rsd.bin"))
            if got == want:
                break
        else:
            sys.exit("could not reproduce the committed index")
        paired = ["-1", os.path.join(td, "reads_1.fq"), "-2", os.path.join(td, "reads_2.fq")]
        indel = ["-1", os.path.join(td, "reads_indel_1.fq"), "-2", os.path.join(td, "reads_indel_2.fq")]
        runs = {
            # built, but without a reference vector so far
            "noDovetail": (paired, ["--noDovetail"]),
            "single": (["-r", os.path.join(td, "reads_1.fq")], []),
            "single_m2_noSensitive": (["-r", os.path.join(td, "reads_2.fq")], ["-m", "2", "--noSensitive"]),
            "indel_default": (indel, []),
            "indel_fuzzy": (indel, ["-f"]),
            # not built yet: selective alignment and chaining
            "selAln": (paired, ["-s"]),
            "selAln_hardFilter": (paired, ["-s", "--hardFilter"]),
            "selAln_recoverOrphans": (paired, ["-s", "--recoverOrphans"]),
            "selAln_minScoreFrac0.9": (paired, ["-s", "--minScoreFrac", "0.9"]),
            "selAln_noOrphans_noDovetail": (paired, ["-s", "--noOrphans", "--noDovetail"]),
            "mimicBT2": (paired, ["--mimicBT2"]),
            "mimicStrictBT2": (paired, ["--mimicStrictBT2"]),
            "chaining": (paired, ["-c"]),
            "single_selAln": (["-r", os.path.join(td, "reads_1.fq")], ["-s"]),
            "indel_selAln": (indel, ["-s"]),
            "indel_selAln_hardFilter": (indel, ["-s", "--hardFilter"]),
            "indel_mimicBT2": (indel, ["--mimicBT2"]),
            "indel_selAln_maxMMPExtension3": (indel, ["-s", "--maxMMPExtension", "3"]),
        }
        for name, (inp, flags) in runs.items():
            sam = os.path.join(td, name + ".sam")
            r = subprocess.run([ref, "quasimap", "-q", "-t", "1", "-i", idx] + inp + ["-o", sam] + flags,
                               stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            if r.returncode != 0 or not os.path.exists(sam):
                print(name, "FAILED", r.stderr[-300:]); continue
            body = b"".join(b"\t".join(l.split(b"\t")[:9] + l.split(b"\t")[10:])
                            for l in open(sam, "rb") if not l.startswith(b"@"))
            with gzip.open(os.path.join(OUT, "expected_%s.noseq.sam.gz" % name), "wb", compresslevel=9) as g:
                g.write(body)
            print(name, "records", body.count(b"\n"))


if __name__ == "__main__":
    main()
