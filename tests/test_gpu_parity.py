"""Parity of the HIP path (libqmap_mi355.so, called through its C ABI) against the oracle.
Bit-exact: hit offsets, hit records, counters, SA-interval lists.  Run on the MI355X box: -m gpu."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLD, load_oracle
from util import assert_hits_equal, pack

pytestmark = pytest.mark.gpu

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


def _gpu(idx, debug=True, ph_compact=False, pair_kernel=True):
    """debug=True keeps the SA-interval records of a fused call (qm_fetch_intervals) -- and so runs the GENERAL stage-A kernel; the kernels the
    headline is quoted on (the pair kernel, qm_lean_kernel) only run with debug=False: _headline() below"""
    import rapmap_amd as ra
    qi = ra.QuasiIndex(idx)
    return qi, ra.QuasiMapper(qi, 0, debug=debug, ph_compact=ph_compact, pair_kernel=pair_kernel)


HEADLINE_KERNELS = ["pair", "lean"]


def _headline(idx, kernel, ph_compact=False):
    """a mapper whose paired calls run the pair kernel (qm_duo.inl) or qm_lean_kernel alone (QM_CTX_NO_PAIR_KERNEL)"""
    return _gpu(idx, debug=False, ph_compact=ph_compact, pair_kernel=kernel == "pair")


def _check_headline(mp, kernel, n_pairs, expect_deferred=True, fuzzy=False, some_merged=True):
    """the call just made went through the kernel it was meant for, and that kernel decided to leave some reads to the general one"""
    assert mp.stat(3) == 2 * n_pairs, "the lean / pair kernel did not run"
    deferred = mp.stat(4)
    assert (0 < deferred < 2 * n_pairs) if expect_deferred else deferred >= 0, deferred
    if kernel == "pair":
        assert mp.stat(9) == n_pairs
        merged = mp.stat(10)
        assert merged == 0 if fuzzy else (1 if some_merged else 0) <= merged <= n_pairs - (deferred + 1) // 2, (merged, deferred)
    else:
        assert mp.stat(9) == -1
    return deferred


def _cmp_ints(res, offs, ints):
    assert np.array_equal(res.ints_offsets, offs)
    oi = res.ints
    for col, name in ((0, "begin"), (1, "end"), (2, "len"), (3, "query_pos"), (5, "list")):
        assert np.array_equal(oi[:, col], ints[name].astype(np.int32)), name


def test_native_library_is_loaded(lib_built):
    """the extension the tests exercise is the in-tree HIP library, not a fallback"""
    import rapmap_amd as ra
    ra.api.lib()
    maps = open("/proc/self/maps").read()
    assert "libqmap_mi355.so" in maps and "libamdhip64" in maps


def test_sample_data_hits_and_sam_md5(sample_data, oracle_mod):
    import rapmap_amd as ra
    import samfmt as sam
    ix, orc = load_oracle(sample_data["idx"])
    qi, mp = _gpu(sample_data["idx"])
    q1, o1 = pack(sample_data["reads1"]); q2, o2 = pack(sample_data["reads2"])
    res = orc.map_pairs(q1, o1, q2, o2, nthreads=2, want_ints=True)
    gr = mp.map_pairs(q1, o1, q2, o2)
    assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "sample_data")
    assert res.counters == gr.counters
    _cmp_ints(res, *mp.intervals(len(o1) - 1))
    body = "".join(sam.format_pair(sample_data["names1"][i], sample_data["reads1"][i], sample_data["names2"][i],
                                   sample_data["reads2"][i], gr.hits[gr.hit_offsets[i]:gr.hit_offsets[i + 1]],
                                   qi.txp_names, qi.txp_lens) for i in range(len(o1) - 1))
    text = "".join(l for l in (sam.sam_header(qi.txp_names, qi.txp_lens) + body).splitlines(True) if not l.startswith("@PG"))
    want = open(os.path.join(GOLD, "sample_data", "expected_sam_body.md5")).read().strip()
    assert hashlib.md5(text.encode()).hexdigest() == want


@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_synth_small(synth_small, oracle_mod, variant):
    import rapmap_amd as ra
    ix, orc = load_oracle(synth_small["idx"])
    qi, mp = _gpu(synth_small["idx"])
    q1, o1 = pack(synth_small["reads1"]); q2, o2 = pack(synth_small["reads2"])
    oo, go = VARIANTS[variant]
    res = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=4, want_ints=True)
    gr = mp.map_pairs(q1, o1, q2, o2, opts=ra.default_opts(**go))
    assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, variant)
    assert res.counters == gr.counters
    _cmp_ints(res, *mp.intervals(len(o1) - 1))


@pytest.mark.parametrize("kernel", HEADLINE_KERNELS)
@pytest.mark.parametrize("variant", [v for v in sorted(VARIANTS) if "noSensitive" not in v])
def test_synth_small_through_the_headline_kernels(synth_small, oracle_mod, variant, kernel):
    """the adversarial golden reads (N's, lower case, IUPAC, reads shorter than k, reads across a `$`, repeat families, indels, ragged
    lengths) through the kernels the headline is quoted on: what they take must be right, and they must LEAVE the reads they are not
    built for (with debug=True these reads only ever met the general kernel)"""
    import rapmap_amd as ra
    ix, orc = load_oracle(synth_small["idx"])
    qi, mp = _headline(synth_small["idx"], kernel)
    q1, o1 = pack(synth_small["reads1"]); q2, o2 = pack(synth_small["reads2"])
    oo, go = VARIANTS[variant]
    res = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=4)
    gr = mp.map_pairs(q1, o1, q2, o2, opts=ra.default_opts(**go))
    assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "%s through the %s kernel" % (variant, kernel))
    assert res.counters == gr.counters
    deferred = _check_headline(mp, kernel, len(o1) - 1, fuzzy="fuzzy" in variant)
    if variant == "default":
        assert deferred == 519, deferred          # (what the lane emulation of both kernels leaves on this set: tests/emu, QM_EMU_LEAN_STATS)


@pytest.mark.parametrize("kernel", HEADLINE_KERNELS)
@pytest.mark.parametrize("variant", ["default", "noStrictCheck", "z0.9", "selAln", "selAln_noStrict"])
def test_reads_with_N_take_the_N_aware_pass(synth_small, oracle_mod, variant, kernel, monkeypatch):
    """round 6: when enough reads of a batch were left for a character outside A C G T (QM_NPASS_MIN: 2 048, here 1), qm_lean_kernel's
    N-aware edition goes over the queue of what the first pass left before the general kernel does -- k-mers with an N stepped over
    (SACollector.hpp:172-181 with its `<=`, :497-512), MMPs ending at one.  Same hits, same counters; the golden set holds reads with an N
    right behind their first k-mer, all-N mates, IUPAC codes (those stay with the general kernel)."""
    import rapmap_amd as ra
    monkeypatch.setenv("QM_NPASS_MIN", "1")
    ix, orc = load_oracle(synth_small["idx"])
    qi, mp = _headline(synth_small["idx"], kernel)
    q1, o1 = pack(synth_small["reads1"]); q2, o2 = pack(synth_small["reads2"])
    oo, go = {"selAln": ({"selAln": 1}, {"sel_aln": 1}), "selAln_noStrict": ({"selAln": 1, "strictCheck": 0}, {"sel_aln": 1, "strict_check": 0})}.get(variant) or VARIANTS[variant]
    res = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=4)
    gr = mp.map_pairs(q1, o1, q2, o2, opts=ra.default_opts(**go))
    assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "%s, N-aware pass behind the %s kernel" % (variant, kernel))
    assert res.counters == gr.counters
    taken = mp.stat(15)
    assert taken > 300, taken
    if variant == "default":
        assert (taken, mp.stat(4)) == (389, 130), (taken, mp.stat(4))      # (the lane emulation: 519 reads left by the first pass, 130 by the second)
    # single-end: the queue names reads, not mates
    res1 = orc.map_single(q1, o1, opts=oracle_mod.default_opts(**oo), nthreads=4)
    gr1 = mp.map_reads(q1, o1, opts=ra.default_opts(**go))
    assert_hits_equal(res1.hit_offsets, res1.hits, gr1.hit_offsets, gr1.hits, "%s, single-end, N-aware pass" % variant)
    assert res1.counters == gr1.counters and mp.stat(15) > 100
    monkeypatch.setenv("QM_NPASS_MIN", "-1")
    gr2 = mp.map_pairs(q1, o1, q2, o2, opts=ra.default_opts(**go))
    assert_hits_equal(res.hit_offsets, res.hits, gr2.hit_offsets, gr2.hits, "%s, no N-aware pass" % variant)
    assert mp.stat(15) == 0


@pytest.mark.parametrize("kernel", HEADLINE_KERNELS)
def test_sample_data_through_the_headline_kernels(sample_data, oracle_mod, kernel):
    import samfmt as sam
    ix, orc = load_oracle(sample_data["idx"])
    qi, mp = _headline(sample_data["idx"], kernel)
    q1, o1 = pack(sample_data["reads1"]); q2, o2 = pack(sample_data["reads2"])
    res = orc.map_pairs(q1, o1, q2, o2, nthreads=2)
    gr = mp.map_pairs(q1, o1, q2, o2)
    assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "sample_data through the %s kernel" % kernel)
    assert res.counters == gr.counters
    _check_headline(mp, kernel, len(o1) - 1, expect_deferred=False)
    body = "".join(sam.format_pair(sample_data["names1"][i], sample_data["reads1"][i], sample_data["names2"][i],
                                   sample_data["reads2"][i], gr.hits[gr.hit_offsets[i]:gr.hit_offsets[i + 1]],
                                   qi.txp_names, qi.txp_lens) for i in range(len(o1) - 1))
    text = "".join(l for l in (sam.sam_header(qi.txp_names, qi.txp_lens) + body).splitlines(True) if not l.startswith("@PG"))
    want = open(os.path.join(GOLD, "sample_data", "expected_sam_body.md5")).read().strip()
    assert hashlib.md5(text.encode()).hexdigest() == want


@pytest.mark.parametrize("kernel", HEADLINE_KERNELS)
def test_dollar_repeats_and_runs_through_the_headline_kernels(synth_small, repeat_data, runs_data, oracle_mod, kernel):
    """reads across a `$`, the > 200-hit and > 1000-interval repeat families (intervals wider than a wavefront, more suffixes than its lanes,
    hits on both strands, tooManyHits) and windows of k equal bases: every one of them a read these kernels must recognise and leave"""
    import rapmap_amd as ra
    from test_emu_parity import _dollar_reads
    r1, r2 = _dollar_reads(synth_small)
    sets = [("dollar", synth_small["idx"], r1, r2), ("repeats", repeat_data["idx"], repeat_data["reads1"], repeat_data["reads2"]),
            ("runs", runs_data["idx"], runs_data["reads1"], runs_data["reads2"])]
    for name, idx, a, b in sets:
        ix, orc = load_oracle(idx)
        qi, mp = _headline(idx, kernel)
        q1, o1 = pack(a); q2, o2 = pack(b)
        for oo, go in (({}, {}), ({"fuzzy": 1}, {"fuzzy": 1}), ({"maxNumHits": 5, "noDovetail": 1}, {"max_num_hits": 5, "no_dovetail": 1})):
            res = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=4)
            gr = mp.map_pairs(q1, o1, q2, o2, opts=ra.default_opts(**go))
            assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "%s %s through the %s kernel" % (name, oo, kernel))
            assert res.counters == gr.counters
            _check_headline(mp, kernel, len(o1) - 1, expect_deferred=name != "dollar", fuzzy="fuzzy" in oo, some_merged=False)
        mp.close(); qi.close()


def test_dollar_in_reads(synth_small, oracle_mod):
    from test_emu_parity import _dollar_reads
    ix, orc = load_oracle(synth_small["idx"])
    qi, mp = _gpu(synth_small["idx"])
    r1, r2 = _dollar_reads(synth_small)
    q1, o1 = pack(r1); q2, o2 = pack(r2)
    res = orc.map_pairs(q1, o1, q2, o2, nthreads=4, want_ints=True)
    gr = mp.map_pairs(q1, o1, q2, o2)
    assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "dollar")
    assert res.counters == gr.counters
    _cmp_ints(res, *mp.intervals(len(o1) - 1))


def test_single_end(synth_small, oracle_mod):
    ix, orc = load_oracle(synth_small["idx"])
    qi, mp = _gpu(synth_small["idx"], debug=False)
    q, o = pack(synth_small["reads1"] + synth_small["reads2"])
    res = orc.map_single(q, o, nthreads=4)
    gr = mp.map_reads(q, o)
    assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "single-end")
    assert res.counters == gr.counters


def test_long_reads_ns4(synth_small, oracle_mod):
    import gzip
    from rapmap_amd import synth
    txt = gzip.open(os.path.join(GOLD, "synth_small", "txome.fa.gz"), "rt").read().split("\n")
    txps = [np.frombuffer(l.upper().encode(), dtype=np.uint8) for l in txt if l and l[0] != ">"][:300]
    txps = [t for t in txps if t.size >= 600 and not (t == ord("N")).any()]
    s1, s2, off, _ = synth.make_reads(txps, 800, seed=5, read_len=250, err=0.01)
    ix, orc = load_oracle(synth_small["idx"])
    qi, mp = _gpu(synth_small["idx"])
    res = orc.map_pairs(s1, off, s2, off, nthreads=4, want_ints=True)
    gr = mp.map_pairs(s1, off, s2, off)
    assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "ns4")
    _cmp_ints(res, *mp.intervals(len(off) - 1))


def test_edge_batches(synth_small, oracle_mod):
    import rapmap_amd as ra
    ix, orc = load_oracle(synth_small["idx"])
    qi, mp = _gpu(synth_small["idx"], debug=False)
    # empty batch
    z = np.zeros(0, np.uint8); zo = np.zeros(1, np.int64)
    gr = mp.map_pairs(z, zo, z, zo)
    assert gr.n_hits == 0 and list(gr.hit_offsets) == [0] and gr.counters["numReads"] == 0
    # batch of one, and reads that are empty / shorter than k
    for r1, r2 in ((synth_small["reads1"][0], synth_small["reads2"][0]), (b"", b"ACGT"), (b"ACGT" * 7, b"")):
        q1, o1 = pack([r1]); q2, o2 = pack([r2])
        res = orc.map_pairs(q1, o1, q2, o2)
        gr = mp.map_pairs(q1, o1, q2, o2)
        assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "one")
    # a read beyond QM_MAX_LONG_READ_LEN is skipped -- empty result, listed by qm_fetch_skipped -- and the rest of the batch is mapped
    # as if it were not there (round 5; before, the call failed): parity with the oracle on the same batch with nothing in its place
    n0 = 40
    r1 = list(synth_small["reads1"][:n0]); r2 = list(synth_small["reads2"][:n0])
    long1 = bytes(np.asarray(ix.text[:2100]).tobytes()).replace(b"$", b"A")
    r1[7] = long1; r2[19] = long1 + b"ACGT"
    q1, o1 = pack(r1); q2, o2 = pack(r2)
    for go, oo in (({}, {}), ({"sel_aln": 1}, {"selAln": 1}), ({"sensitive": 0}, {"sensitive": 0})):
        gr = mp.map_pairs(q1, o1, q2, o2, opts=ra.default_opts(**go))
        tot, reads, codes = mp.skipped()
        assert tot == 2 and sorted(reads.tolist()) == [2 * 7, 2 * 19 + 1] and set(codes.tolist()) == {1}, (go, tot, reads, codes)
        e1 = list(r1); e2 = list(r2); e1[7] = b""; e2[19] = b""
        p1, po1 = pack(e1); p2, po2 = pack(e2)
        res = orc.map_pairs(p1, po1, p2, po2, opts=oracle_mod.default_opts(**oo))
        assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "skipped reads %s" % (go,))
        assert res.counters == gr.counters
    gr = mp.map_pairs(*pack(list(synth_small["reads1"][:n0])), *pack(list(synth_small["reads2"][:n0])))
    assert mp.skipped()[0] == 0
    q1, o1 = pack([b"A" * 600]); q2, o2 = pack([b"C" * 10])
    assert mp.map_pairs(q1, o1, q2, o2).n_hits == 0          # a 600-character read takes the long-read pass, with -s too,
    assert mp.map_pairs(q1, o1, q2, o2, opts=ra.default_opts(sel_aln=1)).n_hits == 0
    assert mp.map_pairs(q1, o1, q2, o2, opts=ra.default_opts(sel_aln=1, dp_bandwidth=120)).n_hits == 0   # ... whatever the band (round 4)


def test_medium_full_parity_and_properties(synth_medium, oracle_mod):
    """60k pairs against a ~5k-transcript index: full bit-exact parity, then size-independent properties."""
    ix, orc = load_oracle(synth_medium["idx"])
    qi, mp = _gpu(synth_medium["idx"], debug=False)
    o = synth_medium["off"]; q1 = synth_medium["seq1"]; q2 = synth_medium["seq2"]
    res = orc.map_pairs(q1, o, q2, o, nthreads=8)
    gr = mp.map_pairs(q1, o, q2, o)
    assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "medium")
    assert res.counters == gr.counters
    # idempotence
    gr2 = mp.map_pairs(q1, o, q2, o)
    assert np.array_equal(gr.hit_offsets, gr2.hit_offsets) and gr.hits.tobytes() == gr2.hits.tobytes()
    # permutation of the pairs permutes the per-pair hit lists
    n = len(o) - 1
    perm = np.random.default_rng(1).permutation(n)
    L = 100
    p1 = q1.reshape(n, L)[perm].reshape(-1); p2 = q2.reshape(n, L)[perm].reshape(-1)
    gp = mp.map_pairs(p1, o, p2, o)
    cnt = np.diff(gr.hit_offsets); cntp = np.diff(gp.hit_offsets)
    assert np.array_equal(cnt[perm], cntp)
    for j in (0, 1, n // 2, n - 1):
        i = perm[j]
        assert gr.hits[gr.hit_offsets[i]:gr.hit_offsets[i + 1]].tobytes() == gp.hits[gp.hit_offsets[j]:gp.hit_offsets[j + 1]].tobytes()
    assert gp.counters == gr.counters
    # structure: per-pair lists sorted by transcript id, bounded by maxNumHits, counters consistent
    h = gr.hits
    assert gr.counters["totHits"] == gr.n_hits == int(gr.hit_offsets[-1])
    assert cnt.max() <= 200
    unit = np.repeat(np.arange(n), cnt)
    same = unit[1:] == unit[:-1]
    paired = h["is_paired"][1:].astype(bool) & same
    assert (h["tid"][1:][paired] > h["tid"][:-1][paired]).all()
    # swapping the mates mirrors every paired hit
    gs = mp.map_pairs(q2, o, q1, o)
    assert np.array_equal(np.diff(gs.hit_offsets)[cnt > 0] > 0, np.ones((cnt > 0).sum(), bool))
    hp = h[h["is_paired"] == 1]; sp = gs.hits[gs.hits["is_paired"] == 1]
    assert hp.size == sp.size
    assert np.array_equal(hp["tid"], sp["tid"]) and np.array_equal(hp["pos"], sp["mate_pos"])
    assert np.array_equal(hp["mate_pos"], sp["pos"]) and np.array_equal(hp["fwd"], sp["mate_is_fwd"])
    assert np.array_equal(hp["frag_len"], sp["frag_len"])


def test_host_buffers_chunked_upload(synth_medium, oracle_mod, monkeypatch):
    """qm_map_pairs / qm_map_reads on host buffers upload the reads chunk by chunk, every chunk's kernel behind its own
    copy: many small, ragged chunks (QM_HOST_CHUNK) must give what one launch gives -- paired, single-end and -s"""
    import rapmap_amd as ra
    ix, orc = load_oracle(synth_medium["idx"])
    qi, mp = _gpu(synth_medium["idx"], debug=False)
    n = 30000
    o = synth_medium["off"][: n + 1]
    q1 = synth_medium["seq1"][: o[-1]]; q2 = synth_medium["seq2"][: o[-1]]
    res = orc.map_pairs(q1, o, q2, o, nthreads=8)
    for chunk in ("7001", "1", None):
        if chunk == "1":
            m = 300                                            # one unit per launch: keep it short
            oo = o[: m + 1]; a1 = q1[: oo[-1]]; a2 = q2[: oo[-1]]
            monkeypatch.setenv("QM_HOST_CHUNK", chunk)
            gr = mp.map_pairs(a1, oo, a2, oo)
            rs = orc.map_pairs(a1, oo, a2, oo, nthreads=4)
            assert_hits_equal(rs.hit_offsets, rs.hits, gr.hit_offsets, gr.hits, "chunk=1")
            continue
        if chunk:
            monkeypatch.setenv("QM_HOST_CHUNK", chunk)
        else:
            monkeypatch.delenv("QM_HOST_CHUNK", raising=False)
        gr = mp.map_pairs(q1, o, q2, o)
        assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "chunk=%s" % chunk)
        assert res.counters == gr.counters
    monkeypatch.setenv("QM_HOST_CHUNK", "4999")
    gs = mp.map_reads(q1, o)
    rs = orc.map_single(q1, o, nthreads=8)
    assert_hits_equal(rs.hit_offsets, rs.hits, gs.hit_offsets, gs.hits, "single-end chunked")
    m = 12000
    oo = o[: m + 1]; a1 = q1[: oo[-1]]; a2 = q2[: oo[-1]]
    gsel = mp.map_pairs(a1, oo, a2, oo, opts=ra.default_opts(sel_aln=1))
    rsel = orc.map_pairs(a1, oo, a2, oo, opts=oracle_mod.default_opts(selAln=1), nthreads=8)
    assert_hits_equal(rsel.hit_offsets, rsel.hits, gsel.hit_offsets, gsel.hits, "-s chunked")


def _fuzz_reads(text, offsets, n, seed, max_len):
    """fragments of the indexed transcripts with substitutions, indels, N's, lower case, random tails and ragged lengths
    (0 .. max_len): nothing a sequencer would not produce, everything the 100-bp ACGT bench reads never exercise"""
    rng = np.random.default_rng(seed)
    comp = np.zeros(256, np.uint8); comp[:] = ord("N")
    for a, b in zip(b"ACGTacgt", b"TGCAtgca"):
        comp[a] = b
    lens = np.diff(np.append(offsets, text.size)) - 1          # every transcript is followed by '$'
    ok = np.nonzero(lens >= 300)[0]
    r1, r2 = [], []
    for i in range(n):
        t = ok[rng.integers(0, ok.size)]
        fl = int(rng.integers(60, min(400, lens[t])))
        st = int(rng.integers(0, lens[t] - fl + 1))
        frag = text[offsets[t] + st: offsets[t] + st + fl].copy()
        mates = []
        for m in range(2):
            L = int(rng.integers(0, max_len + 1)) if rng.random() < 0.15 else int(rng.integers(max(31, max_len // 3), max_len + 1))
            seq = frag[:L].copy() if m == 0 else comp[frag[::-1][:L]].copy()
            k = rng.random()
            if seq.size and k < 0.5:                           # substitutions
                w = rng.random(seq.size) < 0.015
                seq[w] = rng.choice(np.frombuffer(b"ACGT", np.uint8), int(w.sum()))
            if seq.size > 40 and 0.3 < k < 0.6:                # an indel
                p = int(rng.integers(5, seq.size - 5))
                seq = np.delete(seq, p) if rng.random() < 0.5 else np.insert(seq, p, rng.choice(np.frombuffer(b"ACGT", np.uint8)))
            if seq.size and rng.random() < 0.1:
                seq[rng.integers(0, seq.size)] = ord("N")
            if seq.size and rng.random() < 0.1:
                seq = np.frombuffer(seq.tobytes().lower(), np.uint8).copy()
            if seq.size > 50 and rng.random() < 0.05:          # a random tail (adapter)
                seq[-20:] = rng.choice(np.frombuffer(b"ACGT", np.uint8), 20)
            mates.append(seq[:max_len].tobytes())
        if rng.random() < 0.5:
            mates.reverse()
        r1.append(mates[0]); r2.append(mates[1])
    return r1, r2


@pytest.mark.parametrize("max_len", [120, 250])
def test_fuzz_ragged_reads(synth_medium, oracle_mod, max_len):
    """ragged, dirty reads through both read-length kernels (<= 128 bp, <= 256 bp): default, --noSensitive, fuzzy and -s"""
    import rapmap_amd as ra
    ix, orc = load_oracle(synth_medium["idx"])
    qi, mp = _gpu(synth_medium["idx"], debug=False)
    text, offsets = qi.arrays()
    n = 6000
    r1, r2 = _fuzz_reads(np.asarray(text), np.asarray(offsets, dtype=np.int64), n, 77 + max_len, max_len)
    q1, o1 = pack(r1); q2, o2 = pack(r2)
    for oo, go in (({}, {}), ({"sensitive": 0}, {"sensitive": 0}), ({"fuzzy": 1}, {"fuzzy": 1}), ({"selAln": 1}, {"sel_aln": 1}),
                   ({"selAln": 1, "hardFilter": 1}, {"sel_aln": 1, "hard_filter": 1})):
        res = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=8)
        gr = mp.map_pairs(q1, o1, q2, o2, opts=ra.default_opts(**go))
        assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "fuzz %s len<=%d" % (oo, max_len))
        assert res.counters == gr.counters
    rs = orc.map_single(q2, o2, opts=oracle_mod.default_opts(selAln=1), nthreads=8)
    gs = mp.map_reads(q2, o2, opts=ra.default_opts(sel_aln=1))
    assert_hits_equal(rs.hit_offsets, rs.hits, gs.hit_offsets, gs.hits, "fuzz single-end -s len<=%d" % max_len)


def test_device_resident_inputs(synth_medium, oracle_mod):
    """qm_map_device: reads already in HBM (the path bench.py times)"""
    import torch
    ix, orc = load_oracle(synth_medium["idx"])
    qi, mp = _gpu(synth_medium["idx"], debug=False)
    n = 5000
    o = synth_medium["off"][: n + 1]
    q1 = synth_medium["seq1"][: o[-1]]; q2 = synth_medium["seq2"][: o[-1]]
    pad = np.zeros(8, np.uint8)                                  # the device buffers extend past the last read (qmap_mi355.h)
    d1 = torch.from_numpy(np.concatenate([q1, pad])).cuda(); d2 = torch.from_numpy(np.concatenate([q2, pad])).cuda(); do = torch.from_numpy(o).cuda()
    torch.cuda.synchronize()
    gr = mp.map_device(n, d1.data_ptr(), do.data_ptr(), d2.data_ptr(), do.data_ptr(), 100, fetch=True)
    res = orc.map_pairs(q1, o, q2, o, nthreads=4)
    assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "device-resident")
    assert gr.map_kernel_ms > 0


def test_device_resident_batch_in_parts(synth_medium, oracle_mod, monkeypatch):
    """qm_map_device on a large batch maps it as parts in flight together on helper contexts (map_device_split); forced onto a
    small one: offsets, hits and counters are those of the unsplit call and the oracle's -- default, -s, single-end, and an
    uneven number of parts"""
    import torch
    import rapmap_amd as ra
    ix, orc = load_oracle(synth_medium["idx"])
    qi, mp = _gpu(synth_medium["idx"], debug=False)
    n = 20001
    o = synth_medium["off"][: n + 1]
    q1 = synth_medium["seq1"][: o[-1]]; q2 = synth_medium["seq2"][: o[-1]]
    pad = np.zeros(8, np.uint8)
    d1 = torch.from_numpy(np.concatenate([q1, pad])).cuda(); d2 = torch.from_numpy(np.concatenate([q2, pad])).cuda(); do = torch.from_numpy(o).cuda()
    torch.cuda.synchronize()
    for oo, go in (({}, {}), ({"selAln": 1}, {"sel_aln": 1}), ({"sensitive": 0}, {"sensitive": 0})):
        res = orc.map_pairs(q1, o, q2, o, opts=oracle_mod.default_opts(**oo), nthreads=8)
        monkeypatch.setenv("QM_SPLIT", "1")
        whole = mp.map_device(n, d1.data_ptr(), do.data_ptr(), d2.data_ptr(), do.data_ptr(), 100, opts=ra.default_opts(**go), fetch=True)
        monkeypatch.setenv("QM_SPLIT_MIN", "1000")
        for parts in ("2", "3", "5"):
            monkeypatch.setenv("QM_SPLIT", parts)
            gr = mp.map_device(n, d1.data_ptr(), do.data_ptr(), d2.data_ptr(), do.data_ptr(), 100, opts=ra.default_opts(**go), fetch=True)
            assert np.array_equal(gr.hit_offsets, whole.hit_offsets) and gr.hits.tobytes() == whole.hits.tobytes(), (oo, parts)
            assert gr.counters == whole.counters and gr.map_kernel_ms > 0
        assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "in parts %s" % oo)
        assert res.counters == gr.counters
        monkeypatch.delenv("QM_SPLIT_MIN")
    # single-end
    monkeypatch.setenv("QM_SPLIT_MIN", "1000"); monkeypatch.setenv("QM_SPLIT", "3")
    gs = mp.map_device(n, d1.data_ptr(), do.data_ptr(), 0, 0, 100, fetch=True)
    rs = orc.map_single(q1, o, nthreads=8)
    assert_hits_equal(rs.hit_offsets, rs.hits, gs.hit_offsets, gs.hits, "in parts, single-end")


def test_repeat_families(repeat_data, oracle_mod):
    """lists beyond a lane's private memory (wave fix-up), beyond LDS (global scratch), tooManyHits, maxInterval"""
    ix, orc = load_oracle(repeat_data["idx"])
    qi, mp = _gpu(repeat_data["idx"])
    q1, o1 = pack(repeat_data["reads1"]); q2, o2 = pack(repeat_data["reads2"])
    res = orc.map_pairs(q1, o1, q2, o2, nthreads=4, want_ints=True)
    gr = mp.map_pairs(q1, o1, q2, o2)
    assert res.counters["tooManyHits"] > 0
    assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "repeats")
    assert res.counters == gr.counters
    _cmp_ints(res, *mp.intervals(len(o1) - 1))
    import rapmap_amd as ra
    fres = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(fuzzy=1), nthreads=4)
    fgr = mp.map_pairs(q1, o1, q2, o2, opts=ra.default_opts(fuzzy=1))
    assert_hits_equal(fres.hit_offsets, fres.hits, fgr.hit_offsets, fgr.hits, "repeats-fuzzy")
    assert fres.counters == fgr.counters


@pytest.mark.parametrize("variant", ["default", "noSensitive", "fuzzy", "sel"])
def test_homopolymer_runs(runs_data, oracle_mod, variant):
    """windows of k equal bases at every alignment of the four-characters-per-lane strand setup (its shortcut for reads
    without such a window against the tabulating pass and isHomoPolymer, Kmer.hpp:484-487)"""
    import rapmap_amd as ra
    ix, orc = load_oracle(runs_data["idx"])
    qi, mp = _gpu(runs_data["idx"])
    q1, o1 = pack(runs_data["reads1"]); q2, o2 = pack(runs_data["reads2"])
    oo, go = {"default": ({}, {}), "noSensitive": ({"sensitive": 0}, {"sensitive": 0}), "fuzzy": ({"fuzzy": 1}, {"fuzzy": 1}),
              "sel": ({"selAln": 1}, {"sel_aln": 1})}[variant]
    res = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=4, want_ints=variant != "sel")
    gr = mp.map_pairs(q1, o1, q2, o2, opts=ra.default_opts(**go))
    assert res.counters["peHits"] > 0
    assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "runs-" + variant)
    assert res.counters == gr.counters
    if variant != "sel":
        _cmp_ints(res, *mp.intervals(len(o1) - 1))


@pytest.mark.parametrize("compact", [False, True])
def test_perfect_hash_index(synth_small, synth_small_ph, oracle_mod, compact):
    """config 4: `quasiindex -p` index through the HIP path == dense-index oracle; both device images of the index: the
    default (expanded into the one-sector bucket table after every record was checked through the BooPHF walk) and the
    compact one (QM_CTX_PH_COMPACT: the BooPHF levels walked per lookup, as FrugalBooMap::find does)"""
    ix, orc = load_oracle(synth_small["idx"])
    qi, mp = _gpu(synth_small_ph["idx"], ph_compact=compact)
    assert qi.perfect_hash
    q1, o1 = pack(synth_small_ph["reads1"]); q2, o2 = pack(synth_small_ph["reads2"])
    res = orc.map_pairs(q1, o1, q2, o2, nthreads=4, want_ints=True)
    gr = mp.map_pairs(q1, o1, q2, o2)
    assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "perfect-hash")
    assert res.counters == gr.counters
    _cmp_ints(res, *mp.intervals(len(o1) - 1))
    import rapmap_amd as ra
    for oo, go in (({"sensitive": 0}, {"sensitive": 0}), ({"fuzzy": 1, "strictCheck": 0}, {"fuzzy": 1, "strict_check": 0})):
        r2 = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=4)
        g2 = mp.map_pairs(q1, o1, q2, o2, opts=ra.default_opts(**go))
        assert_hits_equal(r2.hit_offsets, r2.hits, g2.hit_offsets, g2.hits, "perfect-hash %s" % oo)
        assert r2.counters == g2.counters


@pytest.mark.parametrize("case", range(6))
def test_selective_alignment(synth_small, oracle_mod, case):
    """config 5 (-s) through the C ABI: hits incl. alignment scores and counters == oracle, paired and single-end"""
    import rapmap_amd as ra
    from test_emu_parity import SEL_VARIANTS, sel_reads
    which, oo, go = SEL_VARIANTS[case]
    ix, orc = load_oracle(synth_small["idx"])
    qi, mp = _gpu(synth_small["idx"], debug=False)
    s1, s2 = sel_reads(synth_small, which)
    q1, o1 = pack(s1); q2, o2 = pack(s2)
    res = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=4)
    gr = mp.map_pairs(q1, o1, q2, o2, opts=ra.default_opts(**go))
    assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "selAln %s" % oo)
    assert res.counters == gr.counters
    rs = orc.map_single(q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=4)
    gs = mp.map_reads(q2, o2, opts=ra.default_opts(**go))
    assert_hits_equal(rs.hit_offsets, rs.hits, gs.hit_offsets, gs.hits, "selAln single-end %s" % oo)
    assert rs.counters == gs.counters


def test_selective_alignment_in_chunks(synth_small, oracle_mod, monkeypatch):
    """stage B/C of -s in four chunks of units (the plan kernels of the later chunks on a stream of their own under the ksw2
    kernel of the earlier ones): what 10 M-pair batches do, forced onto a small one"""
    import rapmap_amd as ra
    monkeypatch.setenv("QM_SEL_CHUNK_UNITS", "500")
    ix, orc = load_oracle(synth_small["idx"])
    qi, mp = _gpu(synth_small["idx"], debug=False)
    q1, o1 = pack(synth_small["reads1"]); q2, o2 = pack(synth_small["reads2"])
    for oo, go in (({"selAln": 1}, {"sel_aln": 1}), ({"selAln": 1, "hardFilter": 1, "maxNumHits": 5}, {"sel_aln": 1, "hard_filter": 1, "max_num_hits": 5})):
        res = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=4)
        gr = mp.map_pairs(q1, o1, q2, o2, opts=ra.default_opts(**go))
        assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "-s in chunks %s" % oo)
        assert res.counters == gr.counters
    rs = orc.map_single(q1, o1, opts=oracle_mod.default_opts(selAln=1), nthreads=4)
    gs = mp.map_reads(q1, o1, opts=ra.default_opts(sel_aln=1))
    assert_hits_equal(rs.hit_offsets, rs.hits, gs.hit_offsets, gs.hits, "-s in chunks, single-end")


def test_selective_alignment_medium(synth_medium, oracle_mod):
    """-s on 20 k pairs against the ~5 k-transcript index"""
    import rapmap_amd as ra
    ix, orc = load_oracle(synth_medium["idx"])
    qi, mp = _gpu(synth_medium["idx"], debug=False)
    n = 20000
    o = synth_medium["off"][: n + 1]
    q1 = synth_medium["seq1"][: o[-1]]; q2 = synth_medium["seq2"][: o[-1]]
    res = orc.map_pairs(q1, o, q2, o, opts=oracle_mod.default_opts(selAln=1), nthreads=8)
    gr = mp.map_pairs(q1, o, q2, o, opts=ra.default_opts(sel_aln=1))
    assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "selAln medium")
    assert res.counters == gr.counters
    # qm_ctx_stat 6 / 7: alignment questions beyond PERFECT chains, and the ksw2 alignments run for them -- with 1 % substitutions most
    # questions are answered by the alignment cache, the ungapped branch or sel_side_score's two rules
    asked, ran = mp.stat(6), mp.stat(7)
    assert asked > n and 0 < ran < asked / 3, (asked, ran)


def test_selective_alignment_list_kernels_agree(synth_medium, repeat_data, oracle_mod, monkeypatch):
    """-s: the list kernel that shares a wavefront among several reads (qm_selpack.inl) and the one-read-per-wavefront kernel
    (QM_SEL_PACK=0: every read) leave the same words for every read, and the hits are the oracle's either way; the repeat
    families bring reads the packed kernel hands on (hits on both strands, more than 64 suffixes)"""
    import rapmap_amd as ra
    n = 12000
    o = synth_medium["off"][: n + 1]
    sets = [(synth_medium["idx"], synth_medium["seq1"][: o[-1]], o, synth_medium["seq2"][: o[-1]], o)]
    q1, o1 = pack(repeat_data["reads1"]); q2, o2 = pack(repeat_data["reads2"])
    sets.append((repeat_data["idx"], q1, o1, q2, o2))
    for idx, a, oa, b, ob in sets:
        ix, orc = load_oracle(idx)
        qi, mp = _gpu(idx, debug=True)
        res = orc.map_pairs(a, oa, b, ob, opts=oracle_mod.default_opts(selAln=1, consensusSlack=0.35), nthreads=8)
        words = {}
        for knob in ("1", "0"):
            monkeypatch.setenv("QM_SEL_PACK", knob)
            gr = mp.map_pairs_stages(a, oa, b, ob, opts=ra.default_opts(sel_aln=1, consensus_slack=0.35))
            v = mp.fetch_stages()
            words[knob] = (v["list_off"].copy(), v["words"].copy())
            full = mp.map_pairs(a, oa, b, ob, opts=ra.default_opts(sel_aln=1, consensus_slack=0.35))
            assert_hits_equal(res.hit_offsets, res.hits, full.hit_offsets, full.hits, "-s, QM_SEL_PACK=%s" % knob)
            assert res.counters == full.counters
        assert np.array_equal(words["1"][0], words["0"][0]) and np.array_equal(words["1"][1], words["0"][1])
        assert words["1"][1].size > 1000


def _medium_txps(idx, min_len=700, cap=1500):
    import rapmap_amd as ra
    qi = ra.QuasiIndex(idx)
    text, offsets = qi.arrays()
    text = np.asarray(text); offsets = np.asarray(offsets, dtype=np.int64)
    ends = np.append(offsets[1:], text.size)
    return [text[a:b - 1] for a, b in zip(offsets, ends) if b - 1 - a >= min_len][:cap]


@pytest.mark.parametrize("variant", ["default", "noSensitive", "fuzzy", "selAln", "perfectHash", "perfectHash_noSensitive", "perfectHash_selAln",
                                     "perfectHashCompact", "perfectHashCompact_noSensitive", "perfectHashCompact_selAln"])
def test_reads_150bp_take_the_three_slot_kernels(synth_medium, synth_medium_ph, oracle_mod, variant):
    """2 x 150 bp (every read 129..192 bp long): the NS=3 instantiations of stage A -- dense / -p, default / --noSensitive /
    fuzzy / -s -- none of which the 100 bp and 250 bp cases reach"""
    import rapmap_amd as ra
    from rapmap_amd import synth
    ph = variant.startswith("perfectHash")
    ix, orc = load_oracle(synth_medium["idx"])                # the oracle maps on the dense index: same hits by construction
    qi, mp = _gpu((synth_medium_ph if ph else synth_medium)["idx"], debug=False, ph_compact="Compact" in variant)
    assert qi.perfect_hash == ph
    s1, s2, off, _ = synth.make_reads(_medium_txps(synth_medium["idx"]), 6000, seed=150, read_len=150, err=0.01)
    # ragged lengths inside the slot range, so that nothing rides on "exactly 150"
    rng = np.random.default_rng(3)
    lens = rng.integers(129, 151, size=len(off) - 1)
    r1 = [s1[off[i]:off[i] + lens[i]].tobytes() for i in range(len(off) - 1)]
    r2 = [s2[off[i]:off[i] + lens[-1 - i]].tobytes() for i in range(len(off) - 1)]
    q1, o1 = pack(r1); q2, o2 = pack(r2)
    assert 128 < int(np.diff(o1).max()) <= 192 and 128 < int(np.diff(o2).max()) <= 192
    oo, go = {"default": ({}, {}), "noSensitive": ({"sensitive": 0}, {"sensitive": 0}), "fuzzy": ({"fuzzy": 1}, {"fuzzy": 1}),
              "selAln": ({"selAln": 1}, {"sel_aln": 1}), "perfectHash": ({}, {}),
              "perfectHash_noSensitive": ({"sensitive": 0}, {"sensitive": 0}), "perfectHash_selAln": ({"selAln": 1}, {"sel_aln": 1}),
              "perfectHashCompact": ({}, {}), "perfectHashCompact_noSensitive": ({"sensitive": 0}, {"sensitive": 0}),
              "perfectHashCompact_selAln": ({"selAln": 1}, {"sel_aln": 1})}[variant]
    res = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=8)
    gr = mp.map_pairs(q1, o1, q2, o2, opts=ra.default_opts(**go))
    assert res.counters["totHits"] > 3000
    assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "150 bp %s" % variant)
    assert res.counters == gr.counters
    rs = orc.map_single(q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=8)
    gs = mp.map_reads(q2, o2, opts=ra.default_opts(**go))
    assert_hits_equal(rs.hit_offsets, rs.hits, gs.hit_offsets, gs.hits, "150 bp single-end %s" % variant)


@pytest.mark.parametrize("band", [34, 40, 64, 120, 1000, -1])
def test_selective_alignment_wide_bands(synth_small, synth_medium, oracle_mod, band):
    """--dpBandwidth beyond 33 runs the same four-per-wavefront ksw2 kernel on a larger column ring"""
    import rapmap_amd as ra
    from test_emu_parity import sel_reads
    ix, orc = load_oracle(synth_small["idx"])
    qi, mp = _gpu(synth_small["idx"], debug=False)
    s1, s2 = sel_reads(synth_small, "indel")
    q1, o1 = pack(s1); q2, o2 = pack(s2)
    res = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(selAln=1, dpBandwidth=band), nthreads=4)
    gr = mp.map_pairs(q1, o1, q2, o2, opts=ra.default_opts(sel_aln=1, dp_bandwidth=band))
    assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "selAln band %d" % band)
    assert res.counters == gr.counters
    # and on 250 bp reads (the longest alignments: 256 x 276 cells inside the band)
    ix2, orc2 = load_oracle(synth_medium["idx"])
    qi2, mp2 = _gpu(synth_medium["idx"], debug=False)
    from rapmap_amd import synth
    a1, a2, ao, _ = synth.make_reads(_medium_txps(synth_medium["idx"]), 1500, seed=250 + band, read_len=250, err=0.02)
    r2 = orc2.map_pairs(a1, ao, a2, ao, opts=oracle_mod.default_opts(selAln=1, dpBandwidth=band), nthreads=8)
    g2 = mp2.map_pairs(a1, ao, a2, ao, opts=ra.default_opts(sel_aln=1, dp_bandwidth=band))
    assert_hits_equal(r2.hit_offsets, r2.hits, g2.hit_offsets, g2.hits, "selAln 250 bp band %d" % band)


def test_selective_alignment_repeats_take_the_slow_pass(repeat_data, oracle_mod):
    """-s, reads inside a 900-copy repeat: more suffixes per strand than a wave's scratch (QM_SEL_CAP) -- queued and redone
    on scratch sized for them; the batch must not fail and the hits must be the oracle's"""
    import rapmap_amd as ra
    ix, orc = load_oracle(repeat_data["idx"])
    qi, mp = _gpu(repeat_data["idx"], debug=False)
    q1, o1 = pack(repeat_data["reads1"]); q2, o2 = pack(repeat_data["reads2"])
    for oo, go in ((dict(selAln=1), dict(sel_aln=1)), (dict(selAln=1, maxNumHits=5000, hardFilter=1), dict(sel_aln=1, max_num_hits=5000, hard_filter=1)),
                   (dict(selAln=1, maxNumHits=5000, consensusSlack=0.5), dict(sel_aln=1, max_num_hits=5000, consensus_slack=0.5))):
        res = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=4)
        gr = mp.map_pairs(q1, o1, q2, o2, opts=ra.default_opts(**go))
        assert mp.stat(2) > 0, "no read took the slow pass"
        assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "repeats -s %s" % oo)
        assert res.counters == gr.counters
    rs = orc.map_single(q1, o1, opts=oracle_mod.default_opts(selAln=1, maxNumHits=5000), nthreads=4)
    gs = mp.map_reads(q1, o1, opts=ra.default_opts(sel_aln=1, max_num_hits=5000))
    assert_hits_equal(rs.hit_offsets, rs.hits, gs.hit_offsets, gs.hits, "repeats -s single-end")


def test_list_buffer_overflow_relaunches_stage_a(repeat_data, oracle_mod):
    """a batch whose per-read hit lists outgrow the buffer sized for an ordinary batch (every read sits in a 300- or 900-copy
    repeat): stage A flags the overflow, the host grows the buffer and redoes the batch -- same results, and the relaunch
    is visible in qm_ctx_stat"""
    ix, orc = load_oracle(repeat_data["idx"])
    core = [i for i in range(48) if 12 <= i < 24 or 36 <= i < 48]         # F300 and F900 reads of the fixture
    u1 = [repeat_data["reads1"][i] for i in core]; u2 = [repeat_data["reads2"][i] for i in core]
    a1, ao1 = pack(u1); a2, ao2 = pack(u2)
    ref = orc.map_pairs(a1, ao1, a2, ao2, nthreads=4)
    reps = 12000                                                           # 288 000 pairs, ~340 M list words (the default buffer of a batch this size: 270 M)
    q1, o1 = pack(u1 * reps); q2, o2 = pack(u2 * reps)
    qi, mp = _gpu(repeat_data["idx"], debug=False)                        # a fresh context: the list buffer starts at its default size
    gr = mp.map_pairs(q1, o1, q2, o2)
    assert mp.stat(0) >= 1, "the batch fitted: no relaunch was exercised (list words %d)" % mp.stat(1)
    nu = len(core)
    cnt = np.diff(gr.hit_offsets).reshape(reps, nu)
    assert (cnt == np.diff(ref.hit_offsets)[None, :]).all()
    for rep in (0, reps // 2, reps - 1):
        b, e = gr.hit_offsets[rep * nu], gr.hit_offsets[(rep + 1) * nu]
        assert gr.hits[b:e].tobytes() == ref.hits.tobytes()
    assert gr.counters["numReads"] == reps * nu and gr.counters["tooManyHits"] == reps * ref.counters["tooManyHits"]
    # the grown buffer serves the next call without another relaunch
    gr2 = mp.map_pairs(q1, o1, q2, o2)
    assert mp.stat(0) == 0 and gr2.hits.tobytes() == gr.hits.tobytes()


@pytest.mark.parametrize("mode", ["plain", "fuzzy", "chain", "noSensitive"])
def test_stage_entry_points_compose_to_the_fused_path(synth_small, oracle_mod, mode):
    """SACollector::operator(), hitsToMappingsSimple and mergeLeftRightHits[Fuzzy] as device calls of their own
    (qm_collect_reads -> qm_hits_to_mappings -> qm_merge_lists): intervals == the oracle's, and the three chained give what
    the fused pass (qm_map_pairs_stages) gives"""
    import rapmap_amd as ra
    ix, orc = load_oracle(synth_small["idx"])
    qi, mp = _gpu(synth_small["idx"], debug=False)
    q1, o1 = pack(synth_small["reads1"]); q2, o2 = pack(synth_small["reads2"])
    n = len(o1) - 1
    go = {"plain": {}, "fuzzy": {"fuzzy": 1}, "chain": {"sel_aln": 1, "fuzzy": 1}, "noSensitive": {"sensitive": 0}}[mode]
    oo = {"plain": {}, "fuzzy": {"fuzzy": 1}, "chain": {"selAln": 1}, "noSensitive": {"sensitive": 0}}[mode]
    opts = ra.default_opts(**go)
    res = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=4, want_ints=True)
    # (1) the collector alone, mate by mate
    fl, iol, il = mp.collect_reads(q1, o1, opts=opts)
    fr, ior, ir = mp.collect_reads(q2, o2, opts=opts)
    oi = res.ints; ooff = res.ints_offsets
    for u in (0, 1, n // 2, n - 1):
        w = oi[ooff[u]:ooff[u + 1]]
        left = w[w[:, 5] < 2]; right = w[w[:, 5] >= 2]
        g = il[iol[u]:iol[u + 1]]; h = ir[ior[u]:ior[u + 1]]
        assert np.array_equal(left[:, 0], g["begin"]) and np.array_equal(left[:, 1], g["end"]) and np.array_equal(left[:, 3], g["query_pos"].astype(np.int32))
        assert np.array_equal(right[:, 0], h["begin"]) and np.array_equal(right[:, 2], h["len"].astype(np.int32))
    assert int(iol[-1] + ior[-1]) == int(ooff[-1])
    # (2) hits -> mappings from those intervals
    ll, wl = mp.hits_to_mappings(np.diff(o1), iol, il, opts=opts)
    lr, wr = mp.hits_to_mappings(np.diff(o2), ior, ir, opts=opts)
    # (3) the merge
    mr, too = mp.merge_lists(ll, wl, lr, wr, fl, fr, np.diff(o1), np.diff(o2), opts=opts)
    fused = mp.map_pairs_stages(q1, o1, q2, o2, opts=opts)
    assert_hits_equal(fused.hit_offsets, fused.hits, mr.hit_offsets, mr.hits, "staged vs fused (%s)" % mode)
    assert fused.counters["peHits"] == mr.counters["peHits"] and fused.counters["seHits"] == mr.counters["seHits"]
    assert fused.counters["tooManyHits"] == mr.counters["tooManyHits"] == int((too & 1).sum())
    # the fused pass keeps the same per-stage outputs
    fo, fi_ = mp.intervals(n)
    assert int(fo[-1]) == int(ooff[-1])
    flo, fw = mp.read_lists(2 * n)
    assert np.array_equal(np.diff(flo)[0::2], np.diff(ll)) and np.array_equal(np.diff(flo)[1::2], np.diff(lr))
    ftm = np.zeros(n + 1, dtype=np.uint8)
    assert ra.api.lib().qm_fetch_too_many(mp._h, ftm.ctypes.data) == 0
    # ... and qm_fetch_stages brings all of them down at once, compacted on the device: the same arrays, per read
    for pinned in (True, False):
        v = mp.fetch_stages(pinned=pinned)
        assert v["n_units"] == n and v["n_reads"] == 2 * n
        assert np.array_equal(v["list_off"], flo) and np.array_equal(v["words"], fw)
        assert np.array_equal(v["iv_off"][0::2], fo) and v["iv"].tobytes() == fi_.tobytes()
        assert np.array_equal(v["iv_off"][1::2] - v["iv_off"][0:-1:2], np.diff(iol)) and np.array_equal(v["found"][0::2], fl) and np.array_equal(v["found"][1::2], fr)
        assert np.array_equal(v["hit_off"], fused.hit_offsets) and v["hits"].tobytes() == fused.hits.tobytes()
        assert np.array_equal(v["too_many"], ftm[:n])
        mp._arena_cap = 0                                  # (the next round allocates the other kind of arena)
    if mode in ("plain", "noSensitive"):
        # without the caller-level bookkeeping the merge result differs from the driver's only where that bookkeeping acts
        full = mp.map_pairs(q1, o1, q2, o2, opts=opts)
        same = np.diff(full.hit_offsets) == np.diff(fused.hit_offsets)
        assert same.mean() > 0.95
        assert_hits_equal(res.hit_offsets, res.hits, full.hit_offsets, full.hits, "fused driver path")


@pytest.mark.parametrize("L", [300, 500])
def test_reads_longer_than_256_bp(synth_medium, synth_medium_ph, oracle_mod, L):
    """2 x 300 and 2 x 500 bp (merged pairs, long-insert libraries): the eight-slot kernels -- dense / -p, default /
    --noSensitive / -s -- instead of QM_E_TOOLONG; a batch that mixes them with short reads runs on the same kernels"""
    import rapmap_amd as ra
    from rapmap_amd import synth
    ix, orc = load_oracle(synth_medium["idx"])
    txps = _medium_txps(synth_medium["idx"], min_len=1200, cap=800)
    s1, s2, off, _ = synth.make_reads(txps, 1500, seed=3 + L, read_len=L, err=0.015)
    for idx, compact in ((synth_medium["idx"], False), (synth_medium_ph["idx"], True)):
        qi, mp = _gpu(idx, debug=False, ph_compact=compact)
        for oo, go in (({}, {}), ({"sensitive": 0}, {"sensitive": 0}), ({"selAln": 1}, {"sel_aln": 1}), ({"fuzzy": 1}, {"fuzzy": 1})):
            res = orc.map_pairs(s1, off, s2, off, opts=oracle_mod.default_opts(**oo), nthreads=8)
            gr = mp.map_pairs(s1, off, s2, off, opts=ra.default_opts(**go))
            assert res.counters["totHits"] > 700
            assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "%d bp %s" % (L, oo))
            assert res.counters == gr.counters
    # mixed with 100 bp reads
    a1, a2, ao, _ = synth.make_reads(txps, 1000, seed=9, read_len=100, err=0.01)
    r1 = [s1[off[i]:off[i + 1]].tobytes() for i in range(300)] + [a1[ao[i]:ao[i + 1]].tobytes() for i in range(1000)]
    r2 = [s2[off[i]:off[i + 1]].tobytes() for i in range(300)] + [a2[ao[i]:ao[i + 1]].tobytes() for i in range(1000)]
    q1, o1 = pack(r1); q2, o2 = pack(r2)
    qi, mp = _gpu(synth_medium["idx"], debug=False)
    res = orc.map_pairs(q1, o1, q2, o2, nthreads=8)
    gr = mp.map_pairs(q1, o1, q2, o2)
    assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "mixed lengths")


@pytest.mark.parametrize("variant", ["selAln", "selAln_band40", "selAln_band20", "selAln_noSensitive", "selAln_perfectHash", "mimicBT2", "selAln_band120", "selAln_fullband"])
def test_selective_alignment_of_reads_beyond_512_bp(synth_medium, synth_medium_ph, oracle_mod, variant):
    """-s on a batch that mixes 2 x 100 bp pairs with reads of 513 .. 2048 bp (the reference aligns any length): the collector sets
    the long reads aside, a second small launch of the 32-slot chain-scoring collector delivers their intervals, the list kernel
    and the plan are length-blind, and the ksw2 row kernel runs in its long-image editions (register, 64- and 128-slot rings)"""
    import rapmap_amd as ra
    from rapmap_amd import synth
    ph = variant == "selAln_perfectHash"
    idx = (synth_medium_ph if ph else synth_medium)["idx"]
    ix, orc = load_oracle(idx)
    txps = _medium_txps(synth_medium["idx"], min_len=2100, cap=300)
    a1, a2, ao, _ = synth.make_reads(txps, 3000, seed=9, read_len=100, err=0.01)
    r1 = [a1[ao[i]:ao[i + 1]].tobytes() for i in range(3000)]; r2 = [a2[ao[i]:ao[i + 1]].tobytes() for i in range(3000)]
    for L, n, err in ((600, 40, 0.01), (1300, 30, 0.02), (2048, 30, 0.005), (2000, 6, 0.0), (513, 10, 0.01)):
        s1, s2, off, _ = synth.make_reads(txps, n, seed=L, read_len=L, err=err)
        for i in range(n):
            at = (37 * i + L) % len(r1)
            r1.insert(at, s1[off[i]:off[i + 1]].tobytes()); r2.insert(at, s2[off[i]:off[i + 1]].tobytes() if i % 3 else a2[ao[i]:ao[i + 1]].tobytes())
    q1, o1 = pack(r1); q2, o2 = pack(r2)
    if variant == "mimicBT2":
        oopts = oracle_mod.mimic_bt2_opts()
        gopts = ra.default_opts(sel_aln=1, aln_policy=1, no_orphans=1, no_dovetail=1, consensus_slack=0.35, max_num_hits=1000)
    else:
        oo, go = {"selAln": ({"selAln": 1}, {"sel_aln": 1}), "selAln_band40": ({"selAln": 1, "dpBandwidth": 40}, {"sel_aln": 1, "dp_bandwidth": 40}),
                  "selAln_band20": ({"selAln": 1, "dpBandwidth": 20}, {"sel_aln": 1, "dp_bandwidth": 20}),
                  "selAln_noSensitive": ({"selAln": 1, "sensitive": 0}, {"sel_aln": 1, "sensitive": 0}),
                  "selAln_perfectHash": ({"selAln": 1}, {"sel_aln": 1}),
                  # bands beyond 97: the row kernel's blocks in device memory (a ring of 4096 slots); round 3 failed these batches
                  "selAln_band120": ({"selAln": 1, "dpBandwidth": 120}, {"sel_aln": 1, "dp_bandwidth": 120}),
                  "selAln_fullband": ({"selAln": 1, "dpBandwidth": -1}, {"sel_aln": 1, "dp_bandwidth": -1})}[variant]
        oopts = oracle_mod.default_opts(**oo); gopts = ra.default_opts(**go)
    res = orc.map_pairs(q1, o1, q2, o2, opts=oopts, nthreads=8)
    qi, mp = _gpu(idx, debug=False)
    gr = mp.map_pairs(q1, o1, q2, o2, opts=gopts)
    assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "-s long reads, %s" % variant)
    assert res.counters == gr.counters
    long_units = [u for u in range(len(r1)) if len(r1[u]) > 512 or len(r2[u]) > 512]
    assert sum(int(res.hit_offsets[u + 1] - res.hit_offsets[u]) > 0 for u in long_units) > len(long_units) // (4 if variant == "mimicBT2" else 2)   # (no orphans there)
    if variant == "selAln":
        rs = orc.map_single(q1, o1, opts=oopts, nthreads=8)
        gs = mp.map_reads(q1, o1, opts=gopts)
        assert_hits_equal(rs.hit_offsets, rs.hits, gs.hit_offsets, gs.hits, "-s long reads, single-end")
        # a batch of long reads only
        lo1 = [r1[u] for u in long_units]; lo2 = [r2[u] for u in long_units]
        p1, po1 = pack(lo1); p2, po2 = pack(lo2)
        rl = orc.map_pairs(p1, po1, p2, po2, opts=oopts, nthreads=8)
        gl = mp.map_pairs(p1, po1, p2, po2, opts=gopts)
        assert_hits_equal(rl.hit_offsets, rl.hits, gl.hit_offsets, gl.hits, "-s, long reads only")


@pytest.mark.parametrize("variant", ["default", "noSensitive", "fuzzy", "perfectHash", "perfectHashCompact"])
def test_reads_beyond_512_bp_take_the_long_read_pass(synth_medium, synth_medium_ph, oracle_mod, variant):
    """a batch of 2 x 100 bp pairs with reads of 600 .. 2048 bp among them (the reference takes any std::string,
    include/SACollector.hpp:108): the short reads run on the two-slot kernels, the long ones are set aside by that launch and
    mapped by a second, small launch of the 32-slot kernels; hits and counters equal the oracle's.  With -s, and beyond 2048
    characters, the call fails loudly instead."""
    import rapmap_amd as ra
    from rapmap_amd import synth
    ph = variant.startswith("perfectHash")
    idx = (synth_medium_ph if ph else synth_medium)["idx"]
    ix, orc = load_oracle(idx)
    txps = _medium_txps(synth_medium["idx"], min_len=2100, cap=300)
    assert len(txps) > 20
    a1, a2, ao, _ = synth.make_reads(txps, 3000, seed=9, read_len=100, err=0.01)
    r1 = [a1[ao[i]:ao[i + 1]].tobytes() for i in range(3000)]; r2 = [a2[ao[i]:ao[i + 1]].tobytes() for i in range(3000)]
    for L, n, err in ((600, 40, 0.01), (1300, 30, 0.02), (2048, 30, 0.005), (2000, 6, 0.0), (513, 10, 0.01)):
        s1, s2, off, _ = synth.make_reads(txps, n, seed=L, read_len=L, err=err)
        for i in range(n):                                 # long reads land between the short ones, on either mate
            at = (37 * i + L) % len(r1)
            r1.insert(at, s1[off[i]:off[i + 1]].tobytes()); r2.insert(at, s2[off[i]:off[i + 1]].tobytes() if i % 3 else a2[ao[i]:ao[i + 1]].tobytes())
    q1, o1 = pack(r1); q2, o2 = pack(r2)
    oo, go = {"default": ({}, {}), "noSensitive": ({"sensitive": 0}, {"sensitive": 0}), "fuzzy": ({"fuzzy": 1}, {"fuzzy": 1}),
              "perfectHash": ({}, {}), "perfectHashCompact": ({}, {})}[variant]
    res = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=8, want_ints=True)
    qi, mp = _gpu(idx, debug=True, ph_compact=variant == "perfectHashCompact")
    gr = mp.map_pairs(q1, o1, q2, o2, opts=ra.default_opts(**go))
    assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "long reads, %s" % variant)
    assert res.counters == gr.counters
    long_units = [u for u in range(len(r1)) if len(r1[u]) > 512 or len(r2[u]) > 512]
    assert mp.stat(2) == sum((len(r1[u]) > 512) + (len(r2[u]) > 512) for u in long_units)       # QM_STAT_SLOW_READS: the reads of the second launch
    assert sum(int(res.hit_offsets[u + 1] - res.hit_offsets[u]) > 0 for u in long_units) > len(long_units) // 2
    io, ints = mp.intervals(len(r1))                         # the SA-interval lists of the fused pass, long reads included
    assert np.array_equal(io, res.ints_offsets)
    assert np.array_equal(ints["begin"], res.ints[:, 0]) and np.array_equal(ints["len"].astype(np.int32), res.ints[:, 2])
    if variant == "default":
        # the collector as a call of its own, single-end reads, the same reads already on the device
        fl, iol, il = mp.collect_reads(q1, o1)
        left = res.ints[res.ints[:, 5] < 2]
        assert int(iol[-1]) == len(left) and np.array_equal(il["begin"], left[:, 0]) and np.array_equal(il["query_pos"].astype(np.int32), left[:, 3])
        rs = orc.map_single(q1, o1, nthreads=8)
        gs = mp.map_reads(q1, o1)
        assert_hits_equal(rs.hit_offsets, rs.hits, gs.hit_offsets, gs.hits, "long reads, single-end")
        import torch
        d1 = torch.from_numpy(np.concatenate([q1, np.zeros(64, np.uint8)])).cuda(); d2 = torch.from_numpy(np.concatenate([q2, np.zeros(64, np.uint8)])).cuda()
        p1 = torch.from_numpy(o1).cuda(); p2 = torch.from_numpy(o2).cuda()
        gd = mp.map_device(len(r1), d1.data_ptr(), p1.data_ptr(), d2.data_ptr(), p2.data_ptr(), 2048, fetch=True)
        assert_hits_equal(res.hit_offsets, res.hits, gd.hit_offsets, gd.hits, "long reads, device-resident input")
        # -s with the whole matrix as the band (round 3: QM_E_TOOLONG for the batch; now the device-memory edition of the row kernel)
        rf = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(selAln=1, dpBandwidth=-1), nthreads=8)
        gf = mp.map_pairs(q1, o1, q2, o2, opts=ra.default_opts(sel_aln=1, dp_bandwidth=-1))
        assert_hits_equal(rf.hit_offsets, rf.hits, gf.hit_offsets, gf.hits, "long reads, -s, full band")
        # one character too many: that read is skipped and listed, the batch is mapped (round 5: the call failed)
        r1[5] = bytes(txps[0][:2049]); q1, o1 = pack(r1)
        gk = mp.map_pairs(q1, o1, q2, o2)
        tot, reads, codes = mp.skipped()
        assert tot == 1 and reads.tolist() == [10] and codes.tolist() == [1]
        r1[5] = b""; q1, o1 = pack(r1)
        rk = orc.map_pairs(q1, o1, q2, o2, nthreads=8)
        assert_hits_equal(rk.hit_offsets, rk.hits, gk.hit_offsets, gk.hits, "long reads, one skipped")


@pytest.mark.parametrize("variant", ["default", "fuzzy", "selAln", "noSensitive"])
def test_two_bit_packed_reads_map_like_their_characters(synth_small, oracle_mod, variant):
    """qm_map_pairs_packed / qm_map_reads_packed (2-bit packed batches over PCIe, unpacked on the device): the dirty
    reads of synth_small -- N, lower case, IUPAC, U, '$', reads shorter than k, ragged lengths -- give the oracle's hits and
    counters bit for bit, as the plain-character calls do"""
    import rapmap_amd as ra
    ix, orc = load_oracle(synth_small["idx"])
    qi, mp = _gpu(synth_small["idx"], debug=False)
    q1, o1 = pack(synth_small["reads1"]); q2, o2 = pack(synth_small["reads2"])
    oo, go = {"default": ({}, {}), "fuzzy": ({"fuzzy": 1}, {"fuzzy": 1}), "selAln": ({"selAln": 1}, {"sel_aln": 1}),
              "noSensitive": ({"sensitive": 0}, {"sensitive": 0})}[variant]
    res = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=4)
    gr = mp.map_pairs_packed(q1, o1, q2, o2, opts=ra.default_opts(**go))
    assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "packed pairs, %s" % variant)
    assert res.counters == gr.counters
    rs = orc.map_single(q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=4)
    gs = mp.map_reads_packed(q2, o2, opts=ra.default_opts(**go))
    assert_hits_equal(rs.hit_offsets, rs.hits, gs.hit_offsets, gs.hits, "packed single-end, %s" % variant)
    if variant == "default":
        z = np.zeros(0, np.uint8); zo = np.zeros(1, np.int64)
        assert mp.map_pairs_packed(z, zo, z, zo).n_hits == 0                 # an empty batch


def _lean_edge_reads(idx, n=600, seed=5):
    """reads that walk the lean kernel's edges: every length from below k to 128 (a 128-character perfect match runs past the SaExt
    window), lower case, N's, long runs of one base, two errors, reverse-complemented mates, empty reads"""
    import rapmap_amd as ra
    rng = np.random.default_rng(seed)
    qi = ra.QuasiIndex(idx)
    text, offsets = qi.arrays()
    text = np.asarray(text); offsets = np.asarray(offsets, dtype=np.int64)
    ends = np.append(offsets[1:], text.size) - 1
    qi.close()
    comp = np.zeros(256, np.uint8); comp[list(b"ACGT")] = list(b"TGCA")
    ok = np.nonzero(ends - offsets >= 300)[0]
    r1, r2 = [], []
    lens = [0, 5, 30, 31, 32, 33, 63, 64, 65, 99, 100, 101, 126, 127, 128]
    for i in range(n):
        t = ok[rng.integers(0, ok.size)]
        L = lens[i % len(lens)] if i % 3 else 100
        a = int(offsets[t] + rng.integers(0, ends[t] - offsets[t] - 260))
        f = text[a:a + L].copy(); m = comp[text[a + 150:a + 150 + L][::-1]].copy()
        kind = (i // len(lens)) % 8
        for r in (f, m):
            if r.size == 0:
                continue
            if kind == 1: r[rng.integers(0, r.size)] = ord("N")
            if kind == 2 and r.size > 40: r[10:10 + 33] = ord("A")
            if kind == 3:
                for _ in range(2): r[rng.integers(0, r.size)] = b"ACGT"[rng.integers(0, 4)]
            if kind == 4: r[:] = np.frombuffer(bytes(r).lower(), dtype=np.uint8)
            if kind == 5 and r.size > 3: r[rng.integers(0, r.size)] = ord("$")
        if kind == 6: f, m = m, f
        r1.append(bytes(f)); r2.append(bytes(m))
    return r1, r2


@pytest.mark.parametrize("variant", ["default", "noStrictCheck", "quasiCov", "fuzzy", "maxInterval3", "noOrphans"])
def test_lean_kernel_edges_match_the_oracle(synth_medium, oracle_mod, variant):
    """the lean stage-A kernel (two reads per wavefront; qm_lean.inl) on reads at its edges, paired and single-end with an odd count:
    what it takes and what it leaves to the general kernel must add up to the oracle's hits"""
    import rapmap_amd as ra
    oo, go = {"default": ({}, {}), "noStrictCheck": ({"strictCheck": 0}, {"strict_check": 0}), "quasiCov": ({"quasiCov": 0.7}, {"quasi_cov": 0.7}),
              "fuzzy": ({"fuzzy": 1}, {"fuzzy": 1}), "maxInterval3": ({"maxInterval": 3}, {"max_interval": 3}),
              "noOrphans": ({"noOrphans": 1}, {"no_orphans": 1})}[variant]
    ix, orc = load_oracle(synth_medium["idx"])
    qi, mp = _gpu(synth_medium["idx"], debug=False)
    r1, r2 = _lean_edge_reads(synth_medium["idx"])
    q1, o1 = pack(r1); q2, o2 = pack(r2)
    res = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=4)
    gr = mp.map_pairs(q1, o1, q2, o2, opts=ra.default_opts(**go))
    assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "lean edges %s" % variant)
    assert res.counters == gr.counters
    assert mp.stat(3) == 2 * len(r1), "the lean kernel was not the one launched"
    assert 0 < mp.stat(4) < 2 * len(r1), "expected some reads left to the general kernel, and most taken"
    one = r1[:-1] + r2[1:2]                                 # single-end, an odd number of reads
    qs, os_ = pack(one)
    if "fuzzy" not in go and "no_orphans" not in go:
        rs = orc.map_single(qs, os_, opts=oracle_mod.default_opts(**oo), nthreads=4)
        gs = mp.map_reads(qs, os_, opts=ra.default_opts(**go))
        assert_hits_equal(rs.hit_offsets, rs.hits, gs.hit_offsets, gs.hits, "lean edges single-end %s" % variant)
        assert mp.stat(3) == len(one)


def test_lean_kernel_on_the_compact_perfect_hash_image(synth_medium, synth_medium_ph, oracle_mod):
    import rapmap_amd as ra
    ix, orc = load_oracle(synth_medium_ph["idx"])
    qi, mp = _gpu(synth_medium_ph["idx"], debug=False, ph_compact=True)
    r1, r2 = _lean_edge_reads(synth_medium["idx"], n=400, seed=6)
    q1, o1 = pack(r1); q2, o2 = pack(r2)
    res = orc.map_pairs(q1, o1, q2, o2, nthreads=4)
    gr = mp.map_pairs(q1, o1, q2, o2)
    assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "lean edges, compact -p")
    assert res.counters == gr.counters and mp.stat(3) == 2 * len(r1)
    gs = mp.map_pairs(q1, o1, q2, o2, opts=ra.default_opts(sel_aln=1))
    rs = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(selAln=1), nthreads=4)
    assert_hits_equal(rs.hit_offsets, rs.hits, gs.hit_offsets, gs.hits, "lean edges, compact -p, -s")


@pytest.mark.gpu
def test_general_kernel_scratch_by_slot(synth_medium, oracle_mod, monkeypatch):
    """the general kernels' device-memory scratch handed out by slot (ReadBatch::gslots: a wave takes a free slot of its XCD when it starts
    and returns it when it ends -- what an oversubscribed launch over a large batch does), forced here for a launch that would not need
    it (QM_GSCR_SLOTS): the interval-keeping stage view of the call surface, whose intervals, lists and hits all equal the oracle's"""
    import rapmap_amd as ra
    monkeypatch.setenv("QM_GSCR_SLOTS", "1")
    ix, orc = load_oracle(synth_medium["idx"])
    qi, mp = _gpu(synth_medium["idx"], debug=True)        # debug: intervals kept -> the general kernel, not the lean one
    n = 30000
    o = synth_medium["off"][: n + 1]
    q1 = synth_medium["seq1"][: o[-1]]; q2 = synth_medium["seq2"][: o[-1]]
    res = orc.map_pairs(q1, o, q2, o, nthreads=8)
    for rep in range(2):                                   # (the second call finds every flag cleared by the first)
        gr = mp.map_pairs(q1, o, q2, o)
        assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "scratch slots, call %d" % rep)
        assert res.counters == gr.counters
        assert mp.stat(3) == -1, "expected the general kernel"


@pytest.mark.gpu
@pytest.mark.parametrize("max_len", [150, 256])
def test_wide_lean_kernel_single_end_and_paired(synth_medium, synth_medium_ph, oracle_mod, max_len):
    """the lean kernel's wide edition (one read of up to 256 characters per wavefront, qm_lean.inl WIDE) on ragged reads with substitutions,
    indels, N's and lower case: paired and single-end (an odd count), lists and the -s collector, dense table and compact -p image --
    what it takes and what it leaves to the general kernel add up to the oracle's hits"""
    import rapmap_amd as ra
    for idx, compact in ((synth_medium["idx"], False), (synth_medium_ph["idx"], True)):
        ix, orc = load_oracle(idx)
        qi = ra.QuasiIndex(idx)
        mp = ra.QuasiMapper(qi, 0, debug=False, ph_compact=compact)
        text, offsets = ra.QuasiIndex(synth_medium["idx"]).arrays()
        r1, r2 = _fuzz_reads(np.asarray(text), np.asarray(offsets, dtype=np.int64), 3001, 4200 + max_len, max_len)
        q1, o1 = pack(r1); q2, o2 = pack(r2)
        for oo, go in (({}, {}), ({"selAln": 1}, {"sel_aln": 1})):
            res = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=8)
            gr = mp.map_pairs(q1, o1, q2, o2, opts=ra.default_opts(**go))
            assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "wide lean, paired %s compact=%s" % (oo, compact))
            assert res.counters == gr.counters
            if not oo:
                assert mp.stat(3) == 2 * len(r1), "the lean kernel was not the one launched"
                assert 0 < mp.stat(4) < len(r1), "expected most reads taken, some left to the general kernel"
            rs = orc.map_single(q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=8)
            gs = mp.map_reads(q2, o2, opts=ra.default_opts(**go))
            assert_hits_equal(rs.hit_offsets, rs.hits, gs.hit_offsets, gs.hits, "wide lean, single-end %s compact=%s" % (oo, compact))
            assert rs.counters == gs.counters
        mp.close()


@pytest.mark.gpu
def test_extension_tables_match_the_byte_per_character_builders(synth_medium, oracle_mod, monkeypatch):
    """round 6 builds SaExt / SaExt2 / sanext out of a 2-bit image of the text (two or three sectors per entry instead of 82); QM_TABLE_CHECK=1
    holds every entry against the byte-per-character builders of rounds 3-5 on the device and fails the build on a difference"""
    import rapmap_amd as ra
    monkeypatch.setenv("QM_TABLE_CHECK", "1")
    ix, orc = load_oracle(synth_medium["idx"])
    qi = ra.QuasiIndex(synth_medium["idx"])
    mp = ra.QuasiMapper(qi, 0, wide_reads=True)            # SaExt with the replica, SaExt2 because asked
    n = 2000
    q1, q2, o = synth_medium["seq1"][: n * 100], synth_medium["seq2"][: n * 100], synth_medium["off"][: n + 1]
    gr = mp.map_pairs(q1, o, q2, o, opts=ra.default_opts(sel_aln=1))       # sanext at the first -s call
    res = orc.map_pairs(q1, o, q2, o, opts=oracle_mod.default_opts(selAln=1), nthreads=4)
    assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "-s with checked tables")
    # the wide table is there (a build that failed its check leaves the replica without it, and reads of 150 characters with the general kernel)
    from rapmap_amd import synth
    s1, s2, off, _ = synth.make_reads(synth_medium["txps"], 500, seed=77, read_len=150, err=0.01)
    gw = mp.map_pairs(s1, off, s2, off)
    rw = orc.map_pairs(s1, off, s2, off, nthreads=4)
    assert_hits_equal(rw.hit_offsets, rw.hits, gw.hit_offsets, gw.hits, "150 bp with checked tables")
    assert mp.stat(3) == 2 * 500, "the wide lean kernel did not run: no SaExt2"
    mp.close(); qi.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["plain", "fuzzy"])
def test_stage_view_without_intervals_and_packed(synth_small, oracle_mod, mode):
    """QM_STAGES_NO_INTERVALS / qm_map_pairs_stages_packed (round 6): everything of the stage view but the interval records -- foundHit, the per-read
    lists, the merge's hits and tooMany flags -- equals the full pass's, with the reads as characters or 2-bit packed; the pass ran on the
    pair / lean kernels (the full one keeps intervals: general kernel)"""
    import rapmap_amd as ra
    qi, mp = _gpu(synth_small["idx"], debug=False)
    q1, o1 = pack(synth_small["reads1"]); q2, o2 = pack(synth_small["reads2"])
    n = len(o1) - 1
    opts = ra.default_opts(**({"fuzzy": 1} if mode == "fuzzy" else {}))
    full = mp.map_pairs_stages(q1, o1, q2, o2, opts=opts)
    vf = {k: (np.array(v, copy=True) if hasattr(v, "shape") else v) for k, v in mp.fetch_stages(pinned=False).items()}
    assert mp.stat(3) == -1
    for packed in (False, True):
        light = mp.map_pairs_stages(q1, o1, q2, o2, opts=opts, no_intervals=True, packed=packed)
        assert mp.stat(3) == 2 * n and mp.stat(4) > 0, "the lean / pair kernel did not run"
        v = mp.fetch_stages(pinned=packed)
        assert np.array_equal(light.hit_offsets, full.hit_offsets) and light.hits.tobytes() == full.hits.tobytes() and light.counters == full.counters
        assert int(v["iv_off"][-1]) == 0 and v["iv"].size == 0
        assert np.array_equal(v["found"], vf["found"])
        assert np.array_equal(v["list_off"], vf["list_off"]) and np.array_equal(v["words"], vf["words"])
        assert np.array_equal(v["hit_off"], vf["hit_off"]) and v["hits"].tobytes() == vf["hits"].tobytes()
        assert np.array_equal(v["too_many"], vf["too_many"])
        mp._arena_cap = 0
    mp.close(); qi.close()
