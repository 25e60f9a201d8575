"""The device mapper's SOURCE (rapmap_amd/csrc/qm_mapper.inl), lane-emulated on the CPU (tests/emu),
against the oracle: bit-exact hits, offsets, counters and SA-interval lists.  This is how the wave
algorithm is debugged without a GPU; the same comparisons run against the real HIP path in
test_gpu_parity.py (-m gpu)."""
import os

import numpy as np
import pytest

from conftest import load_oracle
from util import assert_hits_equal, pack

VARIANTS = {
    "default": ({}, {}),
    "noStrictCheck": ({"strictCheck": 0}, {"strict_check": 0}),
    "z0.9": ({"quasiCov": 0.9}, {"quasi_cov": 0.9}),
    "m3": ({"maxNumHits": 3}, {"max_num_hits": 3}),
    "noOrphans": ({"noOrphans": 1}, {"no_orphans": 1}),
    "noDovetail": ({"noDovetail": 1}, {"no_dovetail": 1}),
    "maxInterval50": ({"maxInterval": 50}, {"max_interval": 50}),
    "noSensitive": ({"sensitive": 0}, {"sensitive": 0}),
    "noSensitive_noStrict": ({"sensitive": 0, "strictCheck": 0}, {"sensitive": 0, "strict_check": 0}),
    "noSensitive_z0.8": ({"sensitive": 0, "quasiCov": 0.8}, {"sensitive": 0, "quasi_cov": 0.8}),
    "fuzzy": ({"fuzzy": 1}, {"fuzzy": 1}),
    "fuzzy_noOrphans_m3": ({"fuzzy": 1, "noOrphans": 1, "maxNumHits": 3}, {"fuzzy": 1, "no_orphans": 1, "max_num_hits": 3}),
    "fuzzy_noDovetail": ({"fuzzy": 1, "noDovetail": 1}, {"fuzzy": 1, "no_dovetail": 1}),
    "fuzzy_noSensitive": ({"fuzzy": 1, "sensitive": 0}, {"fuzzy": 1, "sensitive": 0}),
}


def _emu(idx):
    import emu
    ix, orc = load_oracle(idx)
    return ix, orc, emu.Emu(ix), emu


def _cmp_ints(res, er):
    oi, ei = res.ints, er.ints
    assert np.array_equal(res.ints_offsets, er.int_offsets)
    for col, name in ((0, "begin"), (1, "end"), (2, "len"), (3, "query_pos"), (5, "list")):
        assert np.array_equal(oi[:, col], ei[name].astype(np.int32)), name


def test_sample_data(sample_data, oracle_mod):
    ix, orc, em, emu = _emu(sample_data["idx"])
    q1, o1 = pack(sample_data["reads1"]); q2, o2 = pack(sample_data["reads2"])
    res = orc.map_pairs(q1, o1, q2, o2, nthreads=2, want_ints=True)
    er = em.map(q1, o1, q2, o2)
    assert er.status == 0
    assert_hits_equal(res.hit_offsets, res.hits, er.hit_offsets, er.hits, "sample_data")
    assert res.counters == er.counters
    _cmp_ints(res, er)


@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_synth_small(synth_small, oracle_mod, variant):
    ix, orc, em, emu = _emu(synth_small["idx"])
    q1, o1 = pack(synth_small["reads1"]); q2, o2 = pack(synth_small["reads2"])
    oo, eo = VARIANTS[variant]
    res = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=4, want_ints=True)
    er = em.map(q1, o1, q2, o2, opts=emu.default_opts(**eo))
    assert er.status == 0
    assert_hits_equal(res.hit_offsets, res.hits, er.hit_offsets, er.hits, variant)
    assert res.counters == er.counters
    _cmp_ints(res, er)


def test_crowded_hash_buckets(synth_small, oracle_mod):
    """a table with ~1.4 keys per 2-slot bucket: lookups have to follow the overflow marks into later buckets"""
    import emu
    ix, orc = load_oracle(synth_small["idx"])
    nb = 16
    while nb * 1.4 < ix.hkeys.size:
        nb *= 2
    em = emu.Emu(ix, buckets=nb)
    q1, o1 = pack(synth_small["reads1"]); q2, o2 = pack(synth_small["reads2"])
    res = orc.map_pairs(q1, o1, q2, o2, nthreads=4)
    er = em.map(q1, o1, q2, o2)
    assert_hits_equal(res.hit_offsets, res.hits, er.hit_offsets, er.hits, "crowded")
    assert res.counters == er.counters


def _dollar_reads(sd):
    """reads with a '$' (the text's separator) spliced in, incl. right after the seed k-mer and at the last base"""
    r1 = [bytearray(r) for r in sd["reads1"][:400]]
    r2 = [bytearray(r) for r in sd["reads2"][:400]]
    for i, r in enumerate(r1):
        if len(r) > 40:
            r[(31, 40, len(r) - 1, 5)[i % 4]] = ord("$")
    for i, r in enumerate(r2[::3]):
        if len(r) > 60:
            r[60] = ord("$")
    return [bytes(r) for r in r1], [bytes(r) for r in r2]


def test_dollar_in_reads(synth_small, oracle_mod):
    """'$' in a query sends the MMP extension down the literal binary searches (SASearcher.hpp:154,180)"""
    ix, orc, em, emu = _emu(synth_small["idx"])
    r1, r2 = _dollar_reads(synth_small)
    q1, o1 = pack(r1); q2, o2 = pack(r2)
    res = orc.map_pairs(q1, o1, q2, o2, nthreads=4, want_ints=True)
    er = em.map(q1, o1, q2, o2)
    assert_hits_equal(res.hit_offsets, res.hits, er.hit_offsets, er.hits, "dollar")
    assert res.counters == er.counters
    _cmp_ints(res, er)


def test_single_end(synth_small, oracle_mod):
    ix, orc, em, emu = _emu(synth_small["idx"])
    q, o = pack(synth_small["reads1"] + synth_small["reads2"])
    res = orc.map_single(q, o, nthreads=4)
    er = em.map(q, o)
    assert_hits_equal(res.hit_offsets, res.hits, er.hit_offsets, er.hits, "single-end")
    assert res.counters == er.counters


def test_long_reads_ns4(synth_small, oracle_mod):
    """reads of 129..256 bp take the 4-word (NS=4) instantiation"""
    from rapmap_amd import synth
    import gzip, os
    from conftest import GOLD
    txt = gzip.open(os.path.join(GOLD, "synth_small", "txome.fa.gz"), "rt").read().split("\n")
    txps = [np.frombuffer(l.upper().encode(), dtype=np.uint8) for l in txt if l and l[0] != ">"][:300]
    txps = [t for t in txps if t.size >= 600 and not (t == ord("N")).any()]
    s1, s2, off, _ = synth.make_reads(txps, 800, seed=5, read_len=250, err=0.01)
    a1, a2, aoff, _ = synth.make_reads(txps, 400, seed=6, read_len=151, err=0.02)
    q1 = np.concatenate([s1, a1]); q2 = np.concatenate([s2, a2])
    o = np.concatenate([off, aoff[1:] + off[-1]])
    ix, orc, em, emu = _emu(synth_small["idx"])
    res = orc.map_pairs(q1, o, q2, o, nthreads=4, want_ints=True)
    er = em.map(q1, o, q2, o, ns=4)
    assert er.status == 0 and res.counters["totHits"] > 1000
    assert_hits_equal(res.hit_offsets, res.hits, er.hit_offsets, er.hits, "ns4")
    _cmp_ints(res, er)
    # single-end (the wide lean kernel's other instantiations are cross-checked inside the emulation: every read it takes, word for word)
    rs1 = orc.map_single(q2, o, nthreads=4)
    es1 = em.map(q2, o, ns=4)
    assert es1.status == 0
    assert_hits_equal(rs1.hit_offsets, rs1.hits, es1.hit_offsets, es1.hits, "ns4 single-end")
    rs2 = orc.map_single(q2, o, opts=oracle_mod.default_opts(selAln=1), nthreads=4)
    es2 = em.map(q2, o, opts=emu.default_opts(sel_aln=1), ns=4)
    assert (es2.status & 0xff) == 0
    assert_hits_equal(rs2.hit_offsets, rs2.hits, es2.hit_offsets, es2.hits, "ns4 single-end -s")
    # the NS=2 build must not truncate them silently: they are set aside for the long-read pass, same hits
    er2 = em.map(q1, o, q2, o, ns=2)
    assert (er2.status & 0xff) == 0
    assert_hits_equal(res.hit_offsets, res.hits, er2.hit_offsets, er2.hits, "ns2 + long-read pass")
    # three 64-character slots: the instantiation 2 x 150 bp reads run on (129..192 bp)
    res3 = orc.map_pairs(a1, aoff, a2, aoff, nthreads=4, want_ints=True)
    er3 = em.map(a1, aoff, a2, aoff, ns=3)
    assert er3.status == 0
    assert_hits_equal(res3.hit_offsets, res3.hits, er3.hit_offsets, er3.hits, "ns3")
    _cmp_ints(res3, er3)
    for oo, go in (({"sensitive": 0}, {"sensitive": 0}), ({"selAln": 1}, {"sel_aln": 1})):
        r = orc.map_pairs(a1, aoff, a2, aoff, opts=oracle_mod.default_opts(**oo), nthreads=4)
        e = em.map(a1, aoff, a2, aoff, opts=emu.default_opts(**go), ns=3)
        assert_hits_equal(r.hit_offsets, r.hits, e.hit_offsets, e.hits, "ns3 %s" % oo)
    # the 250 bp reads do not fit three slots: with -s too they are set aside and mapped by the 32-slot kernels
    r = orc.map_pairs(q1, o, q2, o, opts=oracle_mod.default_opts(selAln=1), nthreads=4)
    e = em.map(q1, o, q2, o, opts=emu.default_opts(sel_aln=1), ns=3)
    assert (e.status & 0xff) == 0
    assert_hits_equal(r.hit_offsets, r.hits, e.hit_offsets, e.hits, "-s, reads beyond the slot class")


def test_medium(synth_medium, oracle_mod):
    ix, orc, em, emu = _emu(synth_medium["idx"])
    n = 20000
    o = synth_medium["off"][: n + 1]
    q1 = synth_medium["seq1"][: o[-1]]; q2 = synth_medium["seq2"][: o[-1]]
    res = orc.map_pairs(q1, o, q2, o, nthreads=8, want_ints=True)
    er = em.map(q1, o, q2, o)
    assert_hits_equal(res.hit_offsets, res.hits, er.hit_offsets, er.hits, "medium")
    assert res.counters == er.counters
    _cmp_ints(res, er)


def test_selective_alignment_long_reads_many_suffixes(synth_medium, oracle_mod):
    """-s on 150 and 250 bp reads against the isoform-rich medium index: 15-30 seed intervals per strand bring 45-200
    suffixes, i.e. the lane-parallel sort / chaining over several 64-record chunks (and, beyond 256, lane 0's fallback)"""
    import rapmap_amd as ra
    from rapmap_amd import synth
    ix, orc, em, emu = _emu(synth_medium["idx"])
    qi = ra.QuasiIndex(synth_medium["idx"])
    text, offsets = qi.arrays()
    text = np.asarray(text); offsets = np.asarray(offsets, dtype=np.int64)
    ends = np.append(offsets[1:], text.size)
    txps = [text[a:b - 1] for a, b in zip(offsets, ends) if b - 1 - a >= 700][:1500]
    for L, ns, n in ((150, 3, 1500), (250, 4, 600)):
        s1, s2, off, _ = synth.make_reads(txps, n, seed=11 + L, read_len=L, err=0.01)
        for oo, go in (({"selAln": 1}, {"sel_aln": 1}), ({"selAln": 1, "consensusSlack": 0.35}, {"sel_aln": 1, "consensus_slack": 0.35})):
            res = orc.map_pairs(s1, off, s2, off, opts=oracle_mod.default_opts(**oo), nthreads=8)
            er = em.map(s1, off, s2, off, opts=emu.default_opts(**go), ns=ns)
            assert er.status == 0
            assert_hits_equal(res.hit_offsets, res.hits, er.hit_offsets, er.hits, "-s %d bp %s" % (L, oo))
            assert res.counters == er.counters


def test_repeat_families(repeat_data, oracle_mod):
    """reads inside repeat cores: 40 / 300 / 1100 copies -> lists beyond the LDS lists (global scratch),
    > maxNumHits (tooManyHits) and >= maxInterval (skipped intervals)"""
    ix, orc, em, emu = _emu(repeat_data["idx"])
    q1, o1 = pack(repeat_data["reads1"]); q2, o2 = pack(repeat_data["reads2"])
    res = orc.map_pairs(q1, o1, q2, o2, nthreads=4, want_ints=True)
    er = em.map(q1, o1, q2, o2)
    assert er.status == 0
    assert res.counters["tooManyHits"] > 0 and res.counters["peHits"] > 0
    assert_hits_equal(res.hit_offsets, res.hits, er.hit_offsets, er.hits, "repeats")
    assert res.counters == er.counters
    # fuzzy merge on the same reads (lists keep both orientations; orphans need the mate to have no seed at all)
    fres = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(fuzzy=1), nthreads=4)
    fer = em.map(q1, o1, q2, o2, opts=emu.default_opts(fuzzy=1))
    assert fer.status == 0
    assert_hits_equal(fres.hit_offsets, fres.hits, fer.hit_offsets, fer.hits, "repeats-fuzzy")
    assert fres.counters == fer.counters
    _cmp_ints(res, er)


@pytest.mark.parametrize("variant", ["default", "noSensitive", "fuzzy", "sel"])
def test_homopolymer_runs(runs_data, oracle_mod, variant):
    """windows of k equal bases at every alignment: the setup's shortcut for reads without such a window must agree with
    the tabulating pass and with isHomoPolymer"""
    ix, orc, em, emu = _emu(runs_data["idx"])
    q1, o1 = pack(runs_data["reads1"]); q2, o2 = pack(runs_data["reads2"])
    oo, eo = {"default": ({}, {}), "noSensitive": ({"sensitive": 0}, {"sensitive": 0}), "fuzzy": ({"fuzzy": 1}, {"fuzzy": 1}),
              "sel": ({"selAln": 1}, {"sel_aln": 1})}[variant]
    res = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=4, want_ints=variant != "sel")
    er = em.map(q1, o1, q2, o2, opts=emu.default_opts(**eo))
    assert er.status == 0
    assert res.counters["peHits"] > 0
    assert_hits_equal(res.hit_offsets, res.hits, er.hit_offsets, er.hits, "runs-" + variant)
    assert res.counters == er.counters
    if variant != "sel":
        _cmp_ints(res, er)


def test_perfect_hash_index(synth_small, synth_small_ph, oracle_mod):
    """config 4: BooPHF cascade + FrugalBooMap text verification in the device source; hits must equal the
    dense-index oracle (the reference guarantees the same: SURVEY.md section 4)"""
    ix, orc, em, emu = _emu(synth_small_ph["idx"])
    assert ix.perfect and em.ph
    q1, o1 = pack(synth_small_ph["reads1"]); q2, o2 = pack(synth_small_ph["reads2"])
    res = orc.map_pairs(q1, o1, q2, o2, nthreads=4, want_ints=True)
    er = em.map(q1, o1, q2, o2)
    assert er.status == 0
    assert_hits_equal(res.hit_offsets, res.hits, er.hit_offsets, er.hits, "perfect-hash")
    assert res.counters == er.counters
    _cmp_ints(res, er)
    dix, dorc = load_oracle(synth_small["idx"])
    dres = dorc.map_pairs(q1, o1, q2, o2, nthreads=4)
    assert_hits_equal(dres.hit_offsets, dres.hits, er.hit_offsets, er.hits, "perfect-hash vs dense")
    # the perfect-hash lookups combined with the other compile-time / run-time variants
    for oo, eo in (({"sensitive": 0}, {"sensitive": 0}), ({"fuzzy": 1, "strictCheck": 0}, {"fuzzy": 1, "strict_check": 0})):
        r2 = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=4)
        e2 = em.map(q1, o1, q2, o2, opts=emu.default_opts(**eo))
        assert_hits_equal(r2.hit_offsets, r2.hits, e2.hit_offsets, e2.hits, "perfect-hash %s" % oo)
        assert r2.counters == e2.counters


def test_selective_alignment_collector_intervals(synth_small, oracle_mod):
    """-s, stage A only: chain scoring in the collector (MMPs cut at k + maxMMPExtension, coverage slack 1).  The rest of
    the -s path is not on the device yet, so only the SA-interval hits are compared here."""
    from conftest import GOLD
    import samfmt as sam
    ix, orc, em, emu = _emu(synth_small["idx"])
    n1, s1 = sam.read_fastq(os.path.join(GOLD, "synth_small", "next", "reads_indel_1.fastq.gz"))
    n2, s2 = sam.read_fastq(os.path.join(GOLD, "synth_small", "next", "reads_indel_2.fastq.gz"))
    for r1, r2 in ((synth_small["reads1"], synth_small["reads2"]), (s1, s2)):
        q1, o1 = pack(r1); q2, o2 = pack(r2)
        for ext in (7, 3):
            res = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(selAln=1, maxMMPExtension=ext), nthreads=4, want_ints=True)
            er = em.map(q1, o1, q2, o2, opts=emu.default_opts(sel_aln=1, max_mmp_extension=ext))
            _cmp_ints(res, er)


SEL_VARIANTS = [
    ("reads", dict(selAln=1), dict(sel_aln=1)),
    ("indel", dict(selAln=1), dict(sel_aln=1)),
    ("reads", dict(selAln=1, hardFilter=1), dict(sel_aln=1, hard_filter=1)),
    ("indel", dict(selAln=1, maxMMPExtension=3, minScoreFraction=0.9), dict(sel_aln=1, max_mmp_extension=3, min_score_fraction=0.9)),
    ("reads", dict(selAln=1, noOrphans=1, noDovetail=1, consensusSlack=0.35, maxNumHits=1000, alnPolicy=1),
     dict(sel_aln=1, no_orphans=1, no_dovetail=1, consensus_slack=0.35, max_num_hits=1000, aln_policy=1)),          # --mimicBT2
    ("indel", dict(selAln=1, consensusSlack=0.0, gapOpen=6, gapExtend=3, mismatchPenalty=-6, matchScore=3, dpBandwidth=5),
     dict(sel_aln=1, consensus_slack=0.0, gap_open=6, gap_extend=3, mismatch_penalty=-6, match_score=3, dp_bandwidth=5)),
    # without --strictCheck the other strand's turn depends on the two spot-check counts (hb >= ha): the lean collector's runs of
    # capped MMPs book several hits' counts at once
    ("reads", dict(selAln=1, strictCheck=0), dict(sel_aln=1, strict_check=0)),
    ("indel", dict(selAln=1, strictCheck=0, maxMMPExtension=3), dict(sel_aln=1, strict_check=0, max_mmp_extension=3)),
]


def sel_reads(synth_small, which):
    from conftest import GOLD
    import samfmt as sam
    if which == "reads":
        return synth_small["reads1"], synth_small["reads2"]
    nx = os.path.join(GOLD, "synth_small", "next")
    return sam.read_fastq(os.path.join(nx, "reads_indel_1.fastq.gz"))[1], sam.read_fastq(os.path.join(nx, "reads_indel_2.fastq.gz"))[1]


@pytest.mark.parametrize("case", range(len(SEL_VARIANTS)))
def test_selective_alignment(synth_small, oracle_mod, case):
    """-s end to end in the device source: chaining + multi-position lists (stage A), position-list fuzzy merge, ksw2
    extension alignment, score gate and filter (stage B + C); hits incl. alignment scores and counters == oracle"""
    which, oo, eo = SEL_VARIANTS[case]
    ix, orc, em, emu = _emu(synth_small["idx"])
    s1, s2 = sel_reads(synth_small, which)
    q1, o1 = pack(s1); q2, o2 = pack(s2)
    res = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=4)
    er = em.map(q1, o1, q2, o2, opts=emu.default_opts(**eo))
    assert er.status == 0
    assert_hits_equal(res.hit_offsets, res.hits, er.hit_offsets, er.hits, "selAln %s" % oo)
    assert res.counters == er.counters
    rs = orc.map_single(q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=4)
    es = em.map(q2, o2, opts=emu.default_opts(**eo))
    assert_hits_equal(rs.hit_offsets, rs.hits, es.hit_offsets, es.hits, "selAln single-end %s" % oo)
    assert rs.counters == es.counters


def test_selective_alignment_repeats_take_the_slow_pass(repeat_data, oracle_mod):
    """-s on reads inside repeat cores: the 900-copy family brings more suffixes per strand than a wave's scratch holds
    (QM_SEL_CAP), so those reads are queued and redone on scratch sized for them -- same hits as the oracle, which (like the
    reference) treats them as any other read"""
    ix, orc, em, emu = _emu(repeat_data["idx"])
    q1, o1 = pack(repeat_data["reads1"]); q2, o2 = pack(repeat_data["reads2"])
    for oo, eo in ((dict(selAln=1), dict(sel_aln=1)), (dict(selAln=1, maxNumHits=5000, hardFilter=1), dict(sel_aln=1, max_num_hits=5000, hard_filter=1)),
                   (dict(selAln=1, maxNumHits=5000, consensusSlack=0.5), dict(sel_aln=1, max_num_hits=5000, consensus_slack=0.5))):
        res = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=4)
        er = em.map(q1, o1, q2, o2, opts=emu.default_opts(**eo))
        assert (er.status & 0xff) == 0 and (er.status >> 8) > 0, er.status
        assert_hits_equal(res.hit_offsets, res.hits, er.hit_offsets, er.hits, "repeats -s %s" % oo)
        assert res.counters == er.counters
    rs = orc.map_single(q1, o1, opts=oracle_mod.default_opts(selAln=1, maxNumHits=5000), nthreads=4)
    es = em.map(q1, o1, opts=emu.default_opts(sel_aln=1, max_num_hits=5000))
    assert (es.status >> 8) > 0
    assert_hits_equal(rs.hit_offsets, rs.hits, es.hit_offsets, es.hits, "repeats -s single-end")


@pytest.mark.parametrize("band", [34, 64, 120, -1])
def test_selective_alignment_wide_bands(synth_small, oracle_mod, band):
    """--dpBandwidth beyond 33: the same ksw2 kernel on a larger column ring (128 / 512 slots)"""
    ix, orc, em, emu = _emu(synth_small["idx"])
    s1, s2 = sel_reads(synth_small, "indel")
    q1, o1 = pack(s1); q2, o2 = pack(s2)
    res = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(selAln=1, dpBandwidth=band), nthreads=4)
    er = em.map(q1, o1, q2, o2, opts=emu.default_opts(sel_aln=1, dp_bandwidth=band))
    assert er.status == 0
    assert_hits_equal(res.hit_offsets, res.hits, er.hit_offsets, er.hits, "selAln band %d" % band)
    assert res.counters == er.counters


def test_reads_of_300_and_500_bp_take_the_eight_slot_kernels(synth_medium, oracle_mod):
    """reads longer than 256 bp (merged pairs, long-insert libraries): the NS=8 instantiations, default / --noSensitive / -s"""
    import rapmap_amd as ra
    from rapmap_amd import synth
    ix, orc, em, emu = _emu(synth_medium["idx"])
    qi = ra.QuasiIndex(synth_medium["idx"])
    text, offsets = qi.arrays()
    text = np.asarray(text); offsets = np.asarray(offsets, dtype=np.int64)
    ends = np.append(offsets[1:], text.size)
    txps = [text[a:b - 1] for a, b in zip(offsets, ends) if b - 1 - a >= 1200][:800]
    for L, n in ((300, 500), (500, 250)):
        s1, s2, off, _ = synth.make_reads(txps, n, seed=3 + L, read_len=L, err=0.015)
        for oo, go in (({}, {}), ({"sensitive": 0}, {"sensitive": 0}), ({"selAln": 1}, {"sel_aln": 1})):
            res = orc.map_pairs(s1, off, s2, off, opts=oracle_mod.default_opts(**oo), nthreads=8)
            er = em.map(s1, off, s2, off, opts=emu.default_opts(**go), ns=8)
            assert (er.status & 0xff) == 0, er.status
            assert res.counters["totHits"] > n // 2
            assert_hits_equal(res.hit_offsets, res.hits, er.hit_offsets, er.hits, "%d bp %s" % (L, oo))
            assert res.counters == er.counters
    # 500 bp does not fit four slots: the reads are set aside and mapped by the long-read pass, same hits
    er = em.map(s1, off, s2, off, ns=4)
    res = orc.map_pairs(s1, off, s2, off, nthreads=8)
    assert (er.status & 0xff) == 0, er.status
    assert_hits_equal(res.hit_offsets, res.hits, er.hit_offsets, er.hits, "500 bp through the long-read pass")
    rs = orc.map_pairs(s1, off, s2, off, opts=oracle_mod.default_opts(selAln=1), nthreads=8)
    es = em.map(s1, off, s2, off, opts=emu.default_opts(sel_aln=1), ns=4)
    assert (es.status & 0xff) == 0
    assert_hits_equal(rs.hit_offsets, rs.hits, es.hit_offsets, es.hits, "-s, 500 bp through the long-read pass")


@pytest.mark.parametrize("variant", ["selAln", "selAln_band40", "selAln_noSensitive", "selAln_band120", "selAln_fullband"])
def test_selective_alignment_of_reads_beyond_512_bp(synth_medium, oracle_mod, variant):
    """-s on a batch that mixes 2 x 100 bp pairs with reads of 600 .. 2048 bp (the reference aligns any length): the long reads
    are set aside by the collector, get their intervals from the 32-slot chain-scoring collector, chaining and list assembly are
    length-blind, and the ksw2 row kernel runs with images sized for the longest read (register edition, 64- and 128-slot rings).
    Beyond --dpBandwidth 97 (and with the whole matrix as the band, -1) the blocks of the row kernel live in device memory: a ring
    of 4096 slots holds every column of a 2048-base alignment (round 4; before, such a batch failed with QM_E_TOOLONG)"""
    from rapmap_amd import synth
    import rapmap_amd as ra
    ix, orc, em, emu = _emu(synth_medium["idx"])
    qi = ra.QuasiIndex(synth_medium["idx"])
    text, offsets = qi.arrays()
    text = np.asarray(text); offsets = np.asarray(offsets, dtype=np.int64)
    ends = np.append(offsets[1:], text.size)
    txps = [text[a:b - 1] for a, b in zip(offsets, ends) if b - 1 - a >= 2100][:300]
    a1, a2, ao, _ = synth.make_reads(txps, 200, seed=9, read_len=100, err=0.01)
    r1 = [a1[ao[i]:ao[i + 1]].tobytes() for i in range(200)]; r2 = [a2[ao[i]:ao[i + 1]].tobytes() for i in range(200)]
    for L, n, err in ((600, 8, 0.01), (1300, 6, 0.02), (2048, 6, 0.005), (2000, 3, 0.0), (513, 4, 0.01)):
        s1, s2, off, _ = synth.make_reads(txps, n, seed=L, read_len=L, err=err)
        for i in range(n):
            at = (37 * i + L) % len(r1)
            r1.insert(at, s1[off[i]:off[i + 1]].tobytes()); r2.insert(at, s2[off[i]:off[i + 1]].tobytes() if i % 3 else a2[ao[i]:ao[i + 1]].tobytes())
    q1, o1 = pack(r1); q2, o2 = pack(r2)
    oo, go = {"selAln": ({"selAln": 1}, {"sel_aln": 1}), "selAln_band40": ({"selAln": 1, "dpBandwidth": 40}, {"sel_aln": 1, "dp_bandwidth": 40}),
              "selAln_noSensitive": ({"selAln": 1, "sensitive": 0}, {"sel_aln": 1, "sensitive": 0}),
              "selAln_band120": ({"selAln": 1, "dpBandwidth": 120}, {"sel_aln": 1, "dp_bandwidth": 120}),
              "selAln_fullband": ({"selAln": 1, "dpBandwidth": -1}, {"sel_aln": 1, "dp_bandwidth": -1})}[variant]
    er = em.map(q1, o1, q2, o2, opts=emu.default_opts(**go), ns=2)
    res = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=8)
    assert (er.status & 0xff) == 0, er.status
    assert_hits_equal(res.hit_offsets, res.hits, er.hit_offsets, er.hits, "-s long reads, %s" % variant)
    assert res.counters == er.counters
    long_units = [u for u in range(len(r1)) if len(r1[u]) > 512]
    assert sum(int(res.hit_offsets[u + 1] - res.hit_offsets[u]) > 0 for u in long_units) > len(long_units) // 2


@pytest.mark.parametrize("variant", ["default", "noSensitive", "fuzzy", "perfectHash"])
def test_reads_beyond_512_bp_take_the_long_read_pass(synth_medium, synth_medium_ph, oracle_mod, variant):
    """a batch of 2 x 100 bp pairs with reads of 600 .. 2048 bp among them (the reference takes any std::string,
    include/SACollector.hpp:108): the short reads run on the two-slot kernels, the long ones are set aside and mapped by the
    32-slot kernels (Wide<> flag word, per-read queue); hits, counters and SA-interval lists equal the oracle's"""
    from rapmap_amd import synth
    import rapmap_amd as ra
    idx = (synth_medium_ph if variant == "perfectHash" else synth_medium)["idx"]
    ix, orc, em, emu = _emu(idx)
    qi = ra.QuasiIndex(synth_medium["idx"])
    text, offsets = qi.arrays()
    text = np.asarray(text); offsets = np.asarray(offsets, dtype=np.int64)
    ends = np.append(offsets[1:], text.size)
    txps = [text[a:b - 1] for a, b in zip(offsets, ends) if b - 1 - a >= 2100][:300]
    assert len(txps) > 20
    a1, a2, ao, _ = synth.make_reads(txps, 300, seed=9, read_len=100, err=0.01)
    r1 = [a1[ao[i]:ao[i + 1]].tobytes() for i in range(300)]; r2 = [a2[ao[i]:ao[i + 1]].tobytes() for i in range(300)]
    for L, n, err in ((600, 12, 0.01), (1300, 10, 0.02), (2048, 10, 0.005), (2000, 4, 0.0)):
        s1, s2, off, _ = synth.make_reads(txps, n, seed=L, read_len=L, err=err)
        for i in range(n):                                 # long reads land between the short ones, on either mate
            at = (37 * i + L) % len(r1)
            r1.insert(at, s1[off[i]:off[i + 1]].tobytes()); r2.insert(at, s2[off[i]:off[i + 1]].tobytes() if i % 3 else a2[ao[i]:ao[i + 1]].tobytes())
    q1, o1 = pack(r1); q2, o2 = pack(r2)
    oo, go = {"default": ({}, {}), "noSensitive": ({"sensitive": 0}, {"sensitive": 0}), "fuzzy": ({"fuzzy": 1}, {"fuzzy": 1}), "perfectHash": ({}, {})}[variant]
    res = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=8, want_ints=True)
    er = em.map(q1, o1, q2, o2, opts=emu.default_opts(**go), ns=2)
    assert (er.status & 0xff) == 0, er.status
    assert_hits_equal(res.hit_offsets, res.hits, er.hit_offsets, er.hits, "long reads, %s" % variant)
    assert res.counters == er.counters
    _cmp_ints(res, er)
    long_units = [u for u in range(len(r1)) if len(r1[u]) > 512]
    assert sum(int(res.hit_offsets[u + 1] - res.hit_offsets[u]) > 0 for u in long_units) > len(long_units) // 2
    # one character too many: that read is skipped (empty result, counted), everything else is mapped as before (round 5: the call failed)
    was5 = r1[5]
    r1[5] = bytes(txps[0][:2049]); q1b, o1b = pack(r1)
    er2 = em.map(q1b, o1b, q2, o2, opts=emu.default_opts(**go), ns=2)
    assert (er2.status & 0xff) == 0 and (er2.status >> 24) == 1, er2.status
    r1[5] = b""; q1c, o1c = pack(r1)                        # what the oracle is asked: the same batch with nothing in that read's place
    res2 = orc.map_pairs(q1c, o1c, q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=8)
    assert_hits_equal(res2.hit_offsets, res2.hits, er2.hit_offsets, er2.hits, "skipped read, %s" % variant)
    assert res2.counters == er2.counters
    r1[5] = was5


def test_one_diagonal_chaining_editions_equal_the_doubles():
    """The integer chaining editions of the packed -s list kernels (qm_selpack.inl: sel_chain_diag8 in registers, sel_chain_diag_mem with
    the two previous hits in registers and no walk back on linear chains) against sel_chain_group (the doubles of HitManager.cpp:107-307) on
    random groups of hits on one diagonal: equal query ends, hits without gain, a long first hit, short hits behind long gaps -- the cases
    in which the chain is NOT hit 0 <- hit 1 <- ... included.  Groups on two diagonals must be declined."""
    import ctypes as C
    import emu
    lib = emu._lib()
    lib.qe_chain_diag_check.restype = C.c_int
    lib.qe_chain_diag_check.argtypes = [C.POINTER(C.c_uint32), C.c_int, C.c_int, C.c_int]
    rng = np.random.default_rng(20260930)
    agreed = {8: 0, 0: 0}
    for trial in range(12000):
        read_len = int(rng.choice([75, 100, 150, 250]))
        hn = int(rng.integers(1, 9)) if trial % 2 == 0 else int(rng.integers(1, 40))
        diag = int(rng.integers(0, 5000))
        mode = trial % 5
        ends = []
        e = int(rng.integers(20, 60))
        for i in range(hn):
            if mode == 0:   step = int(rng.integers(0, 3))            # many ties / tiny gains
            elif mode == 1: step = 8                                   # the capped walk of -s
            elif mode == 2: step = int(rng.integers(0, 50))            # gaps longer than a hit
            elif mode == 3: step = int(rng.choice([0, 1, 8, 39, 47]))
            else:           step = int(rng.integers(1, 12))
            e = e + (step if i > 0 else 0)
            ends.append(e)
        recs = np.zeros((hn, 5), np.uint32)
        for i, e in enumerate(ends):
            ln = int(rng.integers(1, 45)) if mode in (0, 2) else int(rng.integers(31, 39))
            if i == 0 and trial % 7 == 0:
                ln = int(rng.integers(31, read_len))                   # the read's first MMP is not capped
            ln = min(ln, e)
            q = e - ln
            recs[i] = (7, diag + q, q, ln, i & 63)
        if trial % 11 == 0 and hn > 1:                                 # an exon the isoform skips: a second diagonal
            k = int(rng.integers(1, hn))
            recs[k:, 1] += 120
        ptr = recs.ctypes.data_as(C.POINTER(C.c_uint32))
        for which in (8, 0):
            r = lib.qe_chain_diag_check(ptr, hn, read_len, which)
            assert r >= 0, "trial %d: edition %d differs from sel_chain_group on %s" % (trial, which, recs.tolist())
            two_diagonals = trial % 11 == 0 and hn > 1
            if two_diagonals or (which == 8 and hn > 8):
                assert r == 0, "trial %d: edition %d took a group it must decline" % (trial, which)
            else:
                assert r == 1, "trial %d: edition %d declined a one-diagonal group" % (trial, which)
                agreed[which] += 1
    assert agreed[8] > 4000 and agreed[0] > 9000


def _edited_pairs(txps, n, seed):
    """pairs of 100-bp reads (mate 2 from the other strand, 150 bases downstream) with the edits that decide between the gapless
    path and a path with one gap: two substitutions anywhere, or a one- / two-base deletion or a one-base insertion a few
    characters from either end of the read (behind it the diagonal has a handful of mismatches, often exactly two)"""
    rng = np.random.default_rng(seed)
    B = np.frombuffer(b"ACGT", dtype=np.uint8)
    comp = np.zeros(256, np.uint8); comp[list(b"ACGT")] = list(b"TGCA")
    long_enough = [t for t in txps if len(t) >= 400]

    def edit(frag):                                         # frag: 104 characters of transcript, the read is (about) its first 100
        kind = int(rng.integers(0, 6))
        r = frag.copy()
        if kind == 0:
            for p in rng.choice(100, 2, replace=False): r[p] = B[(np.searchsorted(B, r[p]) + 1 + rng.integers(0, 3)) % 4]
            return r[:100]
        near = int(rng.integers(1, 7))
        p = near if rng.random() < 0.5 else 100 - near
        if kind in (1, 2): return np.concatenate([r[:p], r[p + 1:]])[:100]                   # one base missing from the read
        if kind == 3: return np.concatenate([r[:p], r[p + 2:]])[:100]                        # two
        if kind == 4: return np.concatenate([r[:p], B[rng.integers(0, 4, 1)], r[p:]])[:100]  # one extra
        for p in rng.choice(100, 3, replace=False): r[p] = B[(np.searchsorted(B, r[p]) + 1) % 4]
        return r[:100]
    r1, r2 = [], []
    for i in range(n):
        t = long_enough[int(rng.integers(0, len(long_enough)))]
        p = int(rng.integers(0, len(t) - 360))
        a = edit(t[p:p + 104]); b = edit(t[p + 150:p + 254])
        b = comp[b[::-1]]
        if i % 2: a, b = b, a
        r1.append(a.tobytes()); r2.append(b.tobytes())
    return r1, r2


@pytest.mark.parametrize("scheme", [dict(), dict(gapOpen=4, gapExtend=2), dict(matchScore=1, mismatchPenalty=-1, gapOpen=1, gapExtend=1),
                                    dict(dpBandwidth=5), dict(matchScore=2, mismatchPenalty=-4, gapOpen=6, gapExtend=1, dpBandwidth=-1)])
def test_selective_alignment_answers_known_without_ksw2(synth_medium, oracle_mod, scheme):
    """sel_side_score's two rules -- the gapless path when it loses no more than one gap, and with two mismatches the best of the
    gapless path and the few one-gap paths that could beat it -- on reads whose edits sit where those paths differ: every
    alignment score (they decide the hits that survive) equals the oracle's, which runs ksw2 for all of them"""
    ix, orc, em, emu = _emu(synth_medium["idx"])
    r1, r2 = _edited_pairs(synth_medium["txps"], 3000, 99)
    q1, o1 = pack(r1); q2, o2 = pack(r2)
    conv = {"gapOpen": "gap_open", "gapExtend": "gap_extend", "matchScore": "match_score", "mismatchPenalty": "mismatch_penalty", "dpBandwidth": "dp_bandwidth"}
    oo = dict(selAln=1, minScoreFraction=0.5, **scheme); eo = dict(sel_aln=1, min_score_fraction=0.5, **{conv[k]: v for k, v in scheme.items()})
    res = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=8)
    er = em.map(q1, o1, q2, o2, opts=emu.default_opts(**eo))
    assert er.status == 0
    assert res.counters["peHits"] > 1000
    assert_hits_equal(res.hit_offsets, res.hits, er.hit_offsets, er.hits, "known answers %s" % scheme)
    assert res.counters == er.counters
