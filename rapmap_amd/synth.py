"""Synthetic GENCODE-like transcriptome and paired-end reads.

There is no network, hence no real GENCODE: genes are 4-12 exons of U[80,400) random ACGT, each gene has
1-9 isoforms (isoform 0 keeps every exon, the others keep each internal exon with p=0.75), reads are
2 x read_len from fragments N(250,25) clipped to [read_len,400], mate 2 reverse-complemented, mates swapped
with p=0.5, i.i.d. substitutions.  Seeded numpy generators; everything ACGT (plus optional N's).
"""
import numpy as np

_B = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
_COMP[list(b"ACGTN")] = list(b"TGCAN")


def make_transcriptome(n_genes, seed=42, paralog_frac=0.0, repeat_family=0):
    """-> (names, list of uint8 arrays).  paralog_frac: that share of the transcripts again as paralogs (4 % substitutions);
    repeat_family: one family of that many transcripts sharing a 300-base core between unique flanks (intervals wider than a wavefront)"""
    rng = np.random.default_rng(seed)
    names, txps = [], []
    for g in range(n_genes):
        nex = int(rng.integers(4, 13))
        lens = rng.integers(80, 400, nex)
        exons = [_B[rng.integers(0, 4, int(l))] for l in lens]
        niso = int(rng.integers(1, 10))
        for i in range(niso):
            keep = rng.random(nex) < 0.75
            keep[0] = True
            keep[-1] = True
            if i == 0:
                keep[:] = True
            txps.append(np.concatenate([e for e, kk in zip(exons, keep) if kk]))
            names.append("G%d.T%d" % (g, i))
    if paralog_frac > 0:
        npar = int(len(txps) * paralog_frac)
        src = rng.integers(0, len(txps), npar)
        for j, s in enumerate(src):
            t = txps[int(s)].copy()
            m = rng.random(t.size) < 0.04
            t[m] = _B[(np.searchsorted(_B, t[m]) + rng.integers(1, 4, int(m.sum()))) % 4]
            txps.append(t)
            names.append("P%d" % j)
    if repeat_family > 0:
        core = _B[rng.integers(0, 4, 300)]
        for j in range(int(repeat_family)):
            txps.append(np.concatenate([_B[rng.integers(0, 4, 250)], core, _B[rng.integers(0, 4, 250)]]))
            names.append("R%d" % j)
    return names, txps


def write_fasta(path, names, txps):
    with open(path, "wb") as f:
        for n, t in zip(names, txps):
            f.write(b">" + n.encode() + b"\n")
            f.write(t.tobytes())
            f.write(b"\n")


def make_reads(txps, n_pairs, seed=43, read_len=100, err=0.01, n_rate=0.0, chunk=1 << 18):
    """-> (seq1 uint8[n*L], seq2 uint8[n*L], offsets int64[n+1], truth (tid, start) int64[n,2])"""
    rng = np.random.default_rng(seed)
    lens = np.array([t.size for t in txps], dtype=np.int64)
    starts = np.zeros(len(txps) + 1, dtype=np.int64)
    np.cumsum(lens, out=starts[1:])
    cat = np.concatenate(txps)
    ok = np.nonzero(lens >= max(400, read_len))[0]
    if ok.size == 0:
        ok = np.nonzero(lens >= read_len)[0]
    L = read_len
    s1 = np.empty(n_pairs * L, dtype=np.uint8)
    s2 = np.empty(n_pairs * L, dtype=np.uint8)
    truth = np.empty((n_pairs, 2), dtype=np.int64)
    ar = np.arange(L, dtype=np.int64)
    for b in range(0, n_pairs, chunk):
        e = min(n_pairs, b + chunk)
        m = e - b
        tid = ok[rng.integers(0, ok.size, m)]
        if L <= 400:
            flen = np.clip(rng.normal(250, 25, m).astype(np.int64), L, 400)
        else:                                                  # long reads: fragments of about two read lengths
            flen = np.clip(rng.normal(2 * L, 40, m).astype(np.int64), L, 3 * L)
        flen = np.minimum(flen, lens[tid])
        st = (rng.random(m) * (lens[tid] - flen + 1)).astype(np.int64)
        g0 = starts[tid] + st
        a = cat[g0[:, None] + ar[None, :]]
        bb = _COMP[cat[(g0 + flen - 1)[:, None] - ar[None, :]]]
        for r in (a, bb):
            msk = rng.random(r.shape) < err
            cnt = int(msk.sum())
            if cnt:
                r[msk] = _B[(np.searchsorted(_B, r[msk]) + rng.integers(1, 4, cnt)) % 4]
            if n_rate > 0:
                r[rng.random(r.shape) < n_rate] = ord("N")
        sw = rng.random(m) < 0.5
        a2 = np.where(sw[:, None], bb, a)
        b2 = np.where(sw[:, None], a, bb)
        s1[b * L:e * L] = a2.reshape(-1)
        s2[b * L:e * L] = b2.reshape(-1)
        truth[b:e, 0] = tid
        truth[b:e, 1] = st
    off = np.arange(n_pairs + 1, dtype=np.int64) * L
    return s1, s2, off, truth


def write_fastq(path, seq, n, L, mate, start=0, append=False):
    """n fixed-width FASTQ records "@r%09d/m" + L bases + "+" + L quality characters, written in one go (record numbers from
    `start`; append=True adds them to an existing file)"""
    rec = np.empty((n, 13), dtype=np.uint8)
    rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
    idx = np.arange(start, start + n, dtype=np.int64)
    for d in range(9):
        rec[:, 10 - d] = ord("0") + (idx // (10 ** d)) % 10
    rec[:, 11] = ord("/"); rec[:, 12] = ord(str(mate))
    body = np.empty((n, 1 + L + 3 + L + 1), dtype=np.uint8)
    body[:, 0] = ord("\n")
    body[:, 1:1 + L] = np.asarray(seq).reshape(n, L)
    body[:, 1 + L] = ord("\n"); body[:, 2 + L] = ord("+"); body[:, 3 + L] = ord("\n")
    body[:, 4 + L:4 + 2 * L] = ord("I")
    body[:, 4 + 2 * L] = ord("\n")
    out = np.concatenate([rec, body], axis=1)
    if append:
        with open(path, "ab") as f:
            out.tofile(f)
    else:
        out.tofile(path)
