"""BigSA: the reference's int64 instantiation, RapMapSAIndex<int64_t, ...> (src/RapMapSAMapper.cpp:1209-1240,
src/HitManager.cpp:887-892; written by the indexer when the text passes 2^31 - 2 characters, src/RapMapSAIndexer.cpp:682-683,
711-722,743-765).  On disk: 8-byte suffix array entries, transcript starts, interval bounds (hash.bin records of 24 bytes),
data_ / overflow_ of a -p index.  On the device every offset is an UNSIGNED 32-bit value (qm_mapper.inl, struct Iv), so the
index is narrowed at open and a text of up to 2^32 - 2 characters fits.

The tests here force the int64 form for the small synthetic transcriptome (QM_FORCE_BIGSA) and hold every layer against the
int32 form of the same text: the files, the numpy reader, the oracle compiled with IndexT = int64_t, the lane-emulated device
code, and (-m gpu) the HIP path.  A text that really is beyond 2^31 characters is a builder-run demonstration
(profiles/r03/bigsa_demo.py): too large for the suite."""
import json
import os

import numpy as np
import pytest

from conftest import load_oracle
from util import assert_hits_equal, pack


def test_indexer_writes_the_int64_form(synth_small, synth_small_big):
    from oracle import q5
    a, b = q5.load(synth_small["idx"]), q5.load(synth_small_big["idx"])
    assert not a.big and b.big
    assert json.load(open(os.path.join(synth_small_big["idx"], "header.json")))["value0"]["BigSA"] is True
    assert b.SA.dtype == np.int64 and b.txpOffsets.dtype == np.int64 and b.hlb.dtype == np.int64
    n = a.SA.size
    assert os.path.getsize(os.path.join(synth_small_big["idx"], "sa.bin")) == 8 + 8 * n
    assert np.array_equal(a.SA, b.SA) and np.array_equal(a.txpOffsets, b.txpOffsets) and np.array_equal(a.text, b.text)
    assert np.array_equal(a.rsd, b.rsd) and a.names == b.names and np.array_equal(a.completeLens, b.completeLens)
    # the same k-mer -> interval map in the same table order (slot placement depends on the key alone)
    assert np.array_equal(a.hkeys, b.hkeys) and np.array_equal(a.hlb, b.hlb) and np.array_equal(a.hub, b.hub)


def test_perfect_hash_index_in_the_int64_form(synth_small_ph, synth_small_big_ph):
    from oracle import q5, q5ph
    a, b = q5.load(synth_small_ph["idx"]), q5.load(synth_small_big_ph["idx"])
    assert b.big and b.perfect
    da, la, oa = q5ph.read_val(os.path.join(synth_small_ph["idx"], "hash_info.val"))
    db, lb, ob = q5ph.read_val(os.path.join(synth_small_big_ph["idx"], "hash_info.val"), big=True)
    assert db.dtype == np.int64 and np.array_equal(da, db) and np.array_equal(la, lb) and oa == ob
    assert open(os.path.join(synth_small_ph["idx"], "hash_info.bph"), "rb").read() == \
        open(os.path.join(synth_small_big_ph["idx"], "hash_info.bph"), "rb").read()
    assert np.array_equal(a.hkeys, b.hkeys) and np.array_equal(a.hlb, b.hlb) and np.array_equal(a.hub, b.hub)


def test_overflow_map_of_a_big_perfect_hash_index_loads_in_the_references_container(tmp_path, lib_built):
    """intervals of 255 and more suffixes live in overflow_, an spp::sparse_hash_map<int64_t, int64_t> in the int64 form, whose
    hasher (spp_hash<int64_t>: a 64-bit mix) differs from the identity of the int32 form: the reference's own container, compiled
    in place (oracle/_ref), must find every entry of the file the indexer wrote"""
    from conftest import _build_big
    from oracle import q5ph
    import test_oracle_ref as tor
    ref = tor._ref()
    if ref is None or not hasattr(ref, "ref_spp64_load"):
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    rng = np.random.default_rng(5)
    unit = "".join("ACGT"[i] for i in rng.integers(0, 4, 400))
    fa = tmp_path / "rep.fa"
    with open(fa, "w") as f:
        for t in range(300):          # 300 copies: every k-mer of the unit has an interval of 300 suffixes
            f.write(">t%d\n%s%s\n" % (t, unit, "".join("ACGT"[i] for i in rng.integers(0, 4, 40))))
    idx = str(tmp_path / "idx")
    _build_big(str(fa), idx, threads=2, perfect_hash=True, keep_duplicates=True)
    data, lens, ovf = q5ph.read_val(os.path.join(idx, "hash_info.val"), big=True)
    assert len(ovf) > 300 and (lens == 255).sum() == len(ovf)
    blob = open(os.path.join(idx, "hash_info.val"), "rb").read()
    start = 8 + 8 * data.size + 8 + lens.size
    tor.check_spp64(ref, blob[start:], ovf)


@pytest.mark.parametrize("variant", ["default", "noSensitive", "fuzzy", "sel"])
def test_oracle64_equals_oracle32(synth_small, synth_small_big, oracle_mod, variant):
    oo = {"default": {}, "noSensitive": {"sensitive": 0}, "fuzzy": {"fuzzy": 1}, "sel": {"selAln": 1}}[variant]
    ix, orc = load_oracle(synth_small["idx"])
    ixb, orcb = load_oracle(synth_small_big["idx"])
    assert orcb.lib.qo_index_bytes() == 8 and orc.lib.qo_index_bytes() == 4
    q1, o1 = pack(synth_small["reads1"]); q2, o2 = pack(synth_small["reads2"])
    a = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=4, want_ints=True)
    b = orcb.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=4, want_ints=True)
    assert_hits_equal(a.hit_offsets, a.hits, b.hit_offsets, b.hits, variant)
    assert a.counters == b.counters and a.work == b.work
    assert np.array_equal(a.ints, b.ints) and np.array_equal(a.ints_offsets, b.ints_offsets)


@pytest.mark.parametrize("variant", ["default", "noSensitive", "fuzzy", "sel", "perfectHash"])
def test_device_code_on_the_int64_form(synth_small_big, synth_small_big_ph, oracle_mod, variant):
    """the device mapper's source, lane-emulated (tests/emu), on the narrowed arrays of a BigSA index == the int64 oracle"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "emu"))
    import emu
    from oracle import q5
    oo, eo = {"default": ({}, {}), "noSensitive": ({"sensitive": 0}, {"sensitive": 0}), "fuzzy": ({"fuzzy": 1}, {"fuzzy": 1}),
              "sel": ({"selAln": 1}, {"sel_aln": 1}), "perfectHash": ({}, {})}[variant]
    data = synth_small_big_ph if variant == "perfectHash" else synth_small_big
    ix, orc = load_oracle(synth_small_big["idx"])
    em = emu.Emu(q5.load(data["idx"]))
    q1, o1 = pack(data["reads1"]); q2, o2 = pack(data["reads2"])
    res = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=4)
    er = em.map(q1, o1, q2, o2, opts=emu.default_opts(**eo))
    assert er.status == 0
    assert_hits_equal(res.hit_offsets, res.hits, er.hit_offsets, er.hits, variant)
    assert res.counters == er.counters


def test_loader_opens_the_int64_form(synth_small, synth_small_big, synth_small_big_ph, lib_built):
    """qm_index_open (host only): a BigSA index opens, reports itself as one and describes the same transcriptome"""
    import rapmap_amd as ra
    a, b, c = ra.QuasiIndex(synth_small["idx"]), ra.QuasiIndex(synth_small_big["idx"]), ra.QuasiIndex(synth_small_big_ph["idx"])
    assert not a.big_sa and b.big_sa and c.big_sa and c.perfect_hash
    for x in (b, c):
        assert (x.n_txps, x.text_len, x.n_keys, x.k) == (a.n_txps, a.text_len, a.n_keys, a.k)
        assert x.txp_names == a.txp_names and np.array_equal(x.txp_lens, a.txp_lens)
        assert np.array_equal(x.arrays()[0], a.arrays()[0]) and np.array_equal(x.arrays()[1], a.arrays()[1])


def test_loader_rejects_damaged_int64_files(synth_small_big, tmp_path, lib_built):
    import shutil
    import rapmap_amd as ra
    d = str(tmp_path / "bad")
    shutil.copytree(synth_small_big["idx"], d)
    with open(os.path.join(d, "sa.bin"), "r+b") as f:          # an entry beyond what 32 unsigned bits hold
        f.seek(8 + 8 * 3); f.write(np.array([1 << 33], dtype="<i8").tobytes())
    with pytest.raises(ra.QmError):
        ra.QuasiIndex(d)
    shutil.rmtree(d); shutil.copytree(synth_small_big["idx"], d)
    with open(os.path.join(d, "sa.bin"), "r+b") as f:          # an int32-sized file under a BigSA header
        f.truncate(8 + 4 * ra.QuasiIndex(synth_small_big["idx"]).text_len)
    with pytest.raises(ra.QmError):
        ra.QuasiIndex(d)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["default", "noSensitive", "fuzzy", "m3_noOrphans", "sel", "perfectHash", "perfectHashCompact"])
def test_hip_path_on_the_int64_form(synth_small_big, synth_small_big_ph, oracle_mod, variant):
    import rapmap_amd as ra
    oo, go = {"default": ({}, {}), "noSensitive": ({"sensitive": 0}, {"sensitive": 0}), "fuzzy": ({"fuzzy": 1}, {"fuzzy": 1}),
              "m3_noOrphans": ({"maxNumHits": 3, "noOrphans": 1}, {"max_num_hits": 3, "no_orphans": 1}),
              "sel": ({"selAln": 1}, {"sel_aln": 1}), "perfectHash": ({}, {}), "perfectHashCompact": ({}, {})}[variant]
    data = synth_small_big_ph if variant.startswith("perfectHash") else synth_small_big
    ix, orc = load_oracle(synth_small_big["idx"])
    qi = ra.QuasiIndex(data["idx"])
    assert qi.big_sa
    mp = ra.QuasiMapper(qi, 0, debug=variant != "sel", ph_compact=variant == "perfectHashCompact")
    q1, o1 = pack(data["reads1"]); q2, o2 = pack(data["reads2"])
    res = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=4, want_ints=variant != "sel")
    gr = mp.map_pairs(q1, o1, q2, o2, opts=ra.default_opts(**go))
    assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, variant)
    assert res.counters == gr.counters
    if variant != "sel":
        offs, ints = mp.intervals(len(o1) - 1)
        assert np.array_equal(res.ints_offsets, offs)
        for col, name in ((0, "begin"), (1, "end"), (2, "len"), (3, "query_pos"), (5, "list")):
            assert np.array_equal(res.ints[:, col], ints[name].astype(np.int32)), name
    rs = orc.map_single(q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=4)
    gs = mp.map_reads(q2, o2, opts=ra.default_opts(**go))
    assert_hits_equal(rs.hit_offsets, rs.hits, gs.hit_offsets, gs.hits, variant + " single-end")
