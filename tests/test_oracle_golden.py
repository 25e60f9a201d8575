"""The oracle (CPU restatement) against every known answer that exists for this path:
 * the SURVEY-recorded md5 of the reference's SAM on its own sample_data (config 1),
 * SAM bodies the reference's probe build wrote for tests/golden/synth_small (see make_golden.py)."""
import gzip
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLD, load_oracle
from util import pack, sam_groups_from_gz, strip_seq

VARIANTS = {
    "default": {},
    "noStrictCheck": {"strictCheck": 0},
    "z0.9": {"quasiCov": 0.9},
    "m3": {"maxNumHits": 3},
    "noOrphans": {"noOrphans": 1},
    "noSensitive": {"sensitive": 0},
    "fuzzy": {"fuzzy": 1},
    "fuzzy_noOrphans_m3": {"fuzzy": 1, "noOrphans": 1, "maxNumHits": 3},
}


def test_sample_data_sam_md5(sample_data, oracle_mod):
    import samfmt as sam
    ix, orc = load_oracle(sample_data["idx"])
    q1, o1 = pack(sample_data["reads1"]); q2, o2 = pack(sample_data["reads2"])
    res = orc.map_pairs(q1, o1, q2, o2, nthreads=2)
    assert res.counters["numReads"] == 10000
    assert res.counters["totHits"] == 14253            # 1.4253 hits / read
    body = []
    for i in range(10000):
        h = res.hits[res.hit_offsets[i]:res.hit_offsets[i + 1]]
        body.append(sam.format_pair(sample_data["names1"][i], sample_data["reads1"][i], sample_data["names2"][i],
                                    sample_data["reads2"][i], h, ix.names, ix.txpLens))
    body = "".join(body)
    assert body.count("\n") == 28506
    text = "".join(l for l in (sam.sam_header(ix.names, ix.txpLens) + body).splitlines(True) if not l.startswith("@PG"))
    want = open(os.path.join(GOLD, "sample_data", "expected_sam_body.md5")).read().strip()
    assert hashlib.md5(text.encode()).hexdigest() == want


@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_synth_small_sam(synth_small, oracle_mod, variant):
    import samfmt as sam
    ix, orc = load_oracle(synth_small["idx"])
    q1, o1 = pack(synth_small["reads1"]); q2, o2 = pack(synth_small["reads2"])
    opts = oracle_mod.default_opts(**VARIANTS[variant])
    res = orc.map_pairs(q1, o1, q2, o2, opts=opts, nthreads=4)
    want = sam_groups_from_gz(os.path.join(GOLD, "synth_small", "expected_%s.noseq.sam.gz" % variant))
    bad = []
    for i in range(len(synth_small["reads1"])):
        h = res.hits[res.hit_offsets[i]:res.hit_offsets[i + 1]]
        txt = sam.format_pair(synth_small["names1"][i], synth_small["reads1"][i], synth_small["names2"][i],
                              synth_small["reads2"][i], h, ix.names, ix.txpLens, opts.maxNumHits)
        mine = strip_seq(txt)
        if mine != want[sam._read_name(synth_small["names1"][i])]:
            bad.append(i)
    assert not bad, "pairs whose SAM differs from the reference: %s" % bad[:10]


def test_threads_do_not_change_results(synth_small, oracle_mod):
    ix, orc = load_oracle(synth_small["idx"])
    q1, o1 = pack(synth_small["reads1"]); q2, o2 = pack(synth_small["reads2"])
    a = orc.map_pairs(q1, o1, q2, o2, nthreads=1)
    b = orc.map_pairs(q1, o1, q2, o2, nthreads=5)
    assert np.array_equal(a.hit_offsets, b.hit_offsets) and a.hits.tobytes() == b.hits.tobytes()
    assert a.counters == b.counters


# ---- second batch of reference vectors (tests/golden/make_golden_next.py): single-end, --noDovetail, reads with indels
NEXT = os.path.join(GOLD, "synth_small", "next")
NEXT_PAIRED = {
    "noDovetail": ("reads", {"noDovetail": 1}),
    "indel_default": ("indel", {}),
    "indel_fuzzy": ("indel", {"fuzzy": 1}),
}
NEXT_SINGLE = {
    "single": ("reads_1", {}),
    "single_m2_noSensitive": ("reads_2", {"maxNumHits": 2, "sensitive": 0}),
}


def _next_reads(synth_small, which):
    import samfmt as sam
    if which == "reads":
        return synth_small["names1"], synth_small["reads1"], synth_small["names2"], synth_small["reads2"]
    n1, s1 = sam.read_fastq(os.path.join(NEXT, "reads_indel_1.fastq.gz"))
    n2, s2 = sam.read_fastq(os.path.join(NEXT, "reads_indel_2.fastq.gz"))
    return n1, s1, n2, s2


@pytest.mark.parametrize("variant", sorted(NEXT_PAIRED))
def test_synth_small_next_paired(synth_small, oracle_mod, variant):
    import samfmt as sam
    which, kw = NEXT_PAIRED[variant]
    n1, s1, n2, s2 = _next_reads(synth_small, which)
    ix, orc = load_oracle(synth_small["idx"])
    q1, o1 = pack(s1); q2, o2 = pack(s2)
    opts = oracle_mod.default_opts(**kw)
    res = orc.map_pairs(q1, o1, q2, o2, opts=opts, nthreads=4)
    want = sam_groups_from_gz(os.path.join(NEXT, "expected_%s.noseq.sam.gz" % variant))
    bad, skipped = [], []
    for i in range(len(s1)):
        h = res.hits[res.hit_offsets[i]:res.hit_offsets[i + 1]]
        mine = strip_seq(sam.format_pair(n1[i], s1[i], n2[i], s2[i], h, ix.names, ix.txpLens, opts.maxNumHits))
        w = want[sam._read_name(n1[i])]
        if mine != w:
            # --noDovetail on ORPHAN hits reads an uninitialised matePos in the reference (undefined behaviour,
            # DESIGN.md section 2): groups that hold orphan records on either side are not comparable
            def orphan(lines):
                fl = [int(x.split("\t")[1]) for x in lines]
                return any((f & 0x8) and not (f & 0x4) for f in fl) or any((f & 0x4) and not (f & 0x8) for f in fl)
            if variant == "noDovetail" and (orphan(w) or orphan(mine)):
                skipped.append(i)
                continue
            bad.append(i)
    assert not bad, "pairs whose SAM differs from the reference: %s" % bad[:10]
    assert len(skipped) < 0.05 * len(s1)


@pytest.mark.parametrize("variant", sorted(NEXT_SINGLE))
def test_synth_small_next_single(synth_small, oracle_mod, variant):
    import samfmt as sam
    which, kw = NEXT_SINGLE[variant]
    names = synth_small["names1"] if which == "reads_1" else synth_small["names2"]
    reads = synth_small["reads1"] if which == "reads_1" else synth_small["reads2"]
    ix, orc = load_oracle(synth_small["idx"])
    q, o = pack(reads)
    opts = oracle_mod.default_opts(**kw)
    res = orc.map_single(q, o, opts=opts, nthreads=4)
    want = {}
    with gzip.open(os.path.join(NEXT, "expected_%s.noseq.sam.gz" % variant), "rt") as f:
        for l in f:
            want.setdefault(l.split("\t", 1)[0], []).append(l)
    bad = []
    for i in range(len(reads)):
        h = res.hits[res.hit_offsets[i]:res.hit_offsets[i + 1]]
        mine = strip_seq(sam.format_single(names[i], reads[i], h, ix.names, ix.txpLens))
        nm = names[i].split(" ")[0]
        if mine != want.get(nm, []):
            bad.append(i)
    assert not bad, "reads whose SAM differs from the reference: %s" % bad[:10]


def test_sample_data_fuzzy(sample_data, oracle_mod):
    """config 1 with -f: the reference's SAM (tests/golden/sample_data/expected_fuzzy.noseq.sam.gz)"""
    import samfmt as sam
    ix, orc = load_oracle(sample_data["idx"])
    q1, o1 = pack(sample_data["reads1"]); q2, o2 = pack(sample_data["reads2"])
    res = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(fuzzy=1), nthreads=2)
    want = sam_groups_from_gz(os.path.join(GOLD, "sample_data", "expected_fuzzy.noseq.sam.gz"))
    bad = []
    for i in range(len(sample_data["reads1"])):
        h = res.hits[res.hit_offsets[i]:res.hit_offsets[i + 1]]
        mine = strip_seq(sam.format_pair(sample_data["names1"][i], sample_data["reads1"][i], sample_data["names2"][i],
                                         sample_data["reads2"][i], h, ix.names, ix.txpLens))
        if mine != want[sam._read_name(sample_data["names1"][i])]:
            bad.append(i)
    assert not bad, bad[:10]


# ---- selective alignment (-s): chaining, multi-position hits, ksw2 extension alignment, score gate (row a17)
def _sel(oracle_mod, **kw):
    return oracle_mod.default_opts(selAln=1, **kw)


SEL_PAIRED = {
    "selAln": ("reads", lambda m: _sel(m)),
    "selAln_hardFilter": ("reads", lambda m: _sel(m, hardFilter=1)),
    "selAln_minScoreFrac0.9": ("reads", lambda m: _sel(m, minScoreFraction=0.9)),
    "selAln_noOrphans_noDovetail": ("reads", lambda m: _sel(m, noOrphans=1, noDovetail=1)),
    "mimicBT2": ("reads", lambda m: m.mimic_bt2_opts()),
    "mimicStrictBT2": ("reads", lambda m: m.mimic_bt2_opts(strict=True)),
    "chaining": ("reads", lambda m: m.default_opts()),              # -c without -s changes nothing (RapMapSAMapper.cpp:182)
    "indel_selAln": ("indel", lambda m: _sel(m)),
    "indel_selAln_hardFilter": ("indel", lambda m: _sel(m, hardFilter=1)),
    "indel_mimicBT2": ("indel", lambda m: m.mimic_bt2_opts()),
    "indel_selAln_maxMMPExtension3": ("indel", lambda m: _sel(m, maxMMPExtension=3)),
}


@pytest.mark.parametrize("variant", sorted(SEL_PAIRED))
def test_synth_small_selective_alignment(synth_small, oracle_mod, variant):
    import samfmt as sam
    which, mk = SEL_PAIRED[variant]
    n1, s1, n2, s2 = _next_reads(synth_small, which)
    ix, orc = load_oracle(synth_small["idx"])
    q1, o1 = pack(s1); q2, o2 = pack(s2)
    opts = mk(oracle_mod)
    res = orc.map_pairs(q1, o1, q2, o2, opts=opts, nthreads=4)
    want = sam_groups_from_gz(os.path.join(NEXT, "expected_%s.noseq.sam.gz" % variant))
    bad = [i for i in range(len(s1))
           if strip_seq(sam.format_pair(n1[i], s1[i], n2[i], s2[i], res.hits[res.hit_offsets[i]:res.hit_offsets[i + 1]],
                                        ix.names, ix.txpLens, opts.maxNumHits)) != want[sam._read_name(n1[i])]]
    assert not bad, "pairs whose SAM (incl. AS:i alignment scores) differs from the reference: %s" % bad[:10]


def test_synth_small_selective_alignment_single_end(synth_small, oracle_mod):
    import samfmt as sam
    ix, orc = load_oracle(synth_small["idx"])
    q, o = pack(synth_small["reads1"])
    res = orc.map_single(q, o, opts=_sel(oracle_mod), nthreads=4)
    want = {}
    with gzip.open(os.path.join(NEXT, "expected_single_selAln.noseq.sam.gz"), "rt") as f:
        for l in f:
            want.setdefault(l.split("\t", 1)[0], []).append(l)
    bad = [i for i in range(len(synth_small["reads1"]))
           if strip_seq(sam.format_single(synth_small["names1"][i], synth_small["reads1"][i],
                                          res.hits[res.hit_offsets[i]:res.hit_offsets[i + 1]], ix.names, ix.txpLens))
           != want.get(synth_small["names1"][i].split(" ")[0], [])]
    assert not bad, bad[:10]


@pytest.mark.parametrize("variant", ["selAln", "selAln_hardFilter", "mimicBT2"])
def test_sample_data_selective_alignment(sample_data, oracle_mod, variant):
    """config 5 on the reference's own sample data"""
    import samfmt as sam
    ix, orc = load_oracle(sample_data["idx"])
    q1, o1 = pack(sample_data["reads1"]); q2, o2 = pack(sample_data["reads2"])
    opts = {"selAln": _sel(oracle_mod), "selAln_hardFilter": _sel(oracle_mod, hardFilter=1), "mimicBT2": oracle_mod.mimic_bt2_opts()}[variant]
    res = orc.map_pairs(q1, o1, q2, o2, opts=opts, nthreads=2)
    want = sam_groups_from_gz(os.path.join(GOLD, "sample_data", "expected_%s.noseq.sam.gz" % variant))
    bad = [i for i in range(len(sample_data["reads1"]))
           if strip_seq(sam.format_pair(sample_data["names1"][i], sample_data["reads1"][i], sample_data["names2"][i],
                                        sample_data["reads2"][i], res.hits[res.hit_offsets[i]:res.hit_offsets[i + 1]],
                                        ix.names, ix.txpLens, opts.maxNumHits)) != want[sam._read_name(sample_data["names1"][i])]]
    assert not bad, bad[:10]


def test_vote_ties_disagree_only_over_U(synth_small, oracle_mod):
    """--noSensitive decides the strand by a k-mer vote over kmerScores after std::sort + std::unique on kpos
    (include/SACollector.hpp:289-337, :297-298).  std::sort is not stable, so the reference's answer is well defined only where
    entries with equal kpos carry equal scores.  The restatement uses a stable sort (first entry wins) and COUNTS what it sees.
    Fuzz: chimeric reads (a forward piece + a reverse-complemented piece: both strands are walked, positions are entered from
    both passes), dirtied with IUPAC codes, lower case, N -- and, separately, with U.
      * without U: many ties, none that disagrees (the stable sort changes nothing the reference could do differently);
      * with U: a window holding a U is a partial word in the forward pass and a whole k-mer in the reverse-complement pass
        (reverseRead: U -> A), the entries disagree -- there the reference itself depends on its standard library's sort; the
        restatement (and the device) keep the first entry, and the count of reads where libstdc++'s std::sort would have
        chosen the other strand is reported."""
    import random
    ix, orc = load_oracle(synth_small["idx"])
    rnd = random.Random(5)
    seqs, cur = [], []
    for line in open(synth_small["fasta"]):
        if line.startswith(">"):
            if cur:
                seqs.append("".join(cur)); cur = []
        else:
            cur.append(line.strip())
    seqs.append("".join(cur))
    seqs = [t for t in seqs if len(t) >= 300]
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    rc = lambda t: "".join(comp.get(c, "N") for c in reversed(t))

    def piece(L):
        t = rnd.choice(seqs); a = rnd.randrange(0, len(t) - L)
        return t[a:a + L]

    def make(dirt):
        L1 = rnd.choice([40, 50, 60, 75]); L2 = rnd.choice([40, 50, 60, 75])
        mode = rnd.randrange(4)
        r = piece(L1) + rc(piece(L2)) if mode == 0 else (rc(piece(L1)) + piece(L2) if mode == 1 else (piece(L1 + L2) if mode == 2 else rc(piece(L1 + L2))))
        r = list(r)
        for _ in range(rnd.choice([0, 0, 1, 1, 2, 4])):
            r[rnd.randrange(len(r))] = rnd.choice(dirt)
        if rnd.random() < 0.3:
            a = rnd.randrange(len(r)); b = min(len(r), a + rnd.randrange(1, 30))
            r[a:b] = [c.lower() for c in r[a:b]]
        return "".join(r).encode()

    def run(dirt):
        r1 = [make(dirt) for _ in range(6000)]; r2 = [make(dirt) for _ in range(6000)]
        oracle_mod.kpos_ties()
        for kw in ({"sensitive": 0}, {"sensitive": 0, "fuzzy": 1}, {"sensitive": 0, "maxNumHits": 50}):
            res = orc.map_pairs(*pack(r1), *pack(r2), opts=oracle_mod.default_opts(**kw), nthreads=4)
            assert res.counters["numReads"] == 6000
        rs = orc.map_single(*pack(r1), opts=oracle_mod.default_opts(sensitive=0), nthreads=4)
        assert rs.counters["numReads"] == 6000
        return oracle_mod.kpos_ties()
    clean = run("RYKMSWBDHVNn")
    assert clean["ties"] > 1000 and clean["conflicts"] == 0, clean
    withU = run("RYKMSWBDHVNnUu")
    assert withU["conflicts"] > 0 and withU["conflicts_without_U"] == 0, withU
    print("vote ties with U in the reads:", withU)


def test_perfect_hash_oracle_table_cache(synth_small_ph, tmp_path):
    """q5.load(enum_cache=...): the k-mer table of a -p index (enumerated from the suffix array: q5ph.enumerate_intervals) written by one process and
    read back by another -- bench.py's background child prepares it beside the index it builds -- is the table itself"""
    import numpy as np
    from oracle import q5
    pre = str(tmp_path / "oracle_enum")
    a = q5.load(synth_small_ph["idx"], enum_cache=pre)          # computes, writes
    files = sorted(p.name for p in tmp_path.iterdir())
    assert files == ["oracle_enum_keys.npy", "oracle_enum_lb.npy", "oracle_enum_ub.npy"], files
    b = q5.load(synth_small_ph["idx"], enum_cache=pre)          # reads
    c = q5.load(synth_small_ph["idx"])                          # no cache
    for x, y, z in ((a.hkeys, b.hkeys, c.hkeys), (a.hlb, b.hlb, c.hlb), (a.hub, b.hub, c.hub)):
        assert x.dtype == y.dtype == z.dtype and np.array_equal(x, y) and np.array_equal(x, z)
