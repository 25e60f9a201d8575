"""The oracle's restatements against the REFERENCE ITSELF, for the pieces of the path whose own source files compile in this
image (oracle/Makefile.ref -> oracle/_ref/libqm_ref.so, built from /root/reference in place; see oracle/ref_harness.cpp):
ksw2pp's KSW2Aligner + ksw_extz2_sse (row a17), the Kmer<32,1> codec (a1), boomphf::mphf load + lookup (a3) and rank9b (a8).
This is what pins those four restatements; the collector / searcher / hit manager templates cannot be compiled here (they
include the un-vendored cereal) and stay corroborated by the reference's SAM fixtures only."""
import ctypes as C
import os
import subprocess
import tarfile

import numpy as np
import pytest

from conftest import GOLD, ROOT

REF_SO = os.path.join(ROOT, "oracle", "_ref", "libqm_ref.so")


@pytest.fixture(scope="module")
def ref():
    if not os.path.exists(REF_SO):
        if os.path.isdir("/root/reference/src/ksw2pp"):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "-f", "Makefile.ref"])
        else:
            pytest.skip("oracle/_ref not built and the reference tree is not here")
    L = C.CDLL(REF_SO)
    L.ref_ksw_extension.restype = C.c_int
    L.ref_ksw_extension.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int] + [C.c_int] * 5
    L.ref_kmer.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
    L.ref_mphf_load.restype = C.c_void_p; L.ref_mphf_load.argtypes = [C.c_char_p]
    L.ref_mphf_lookup.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.ref_mphf_free.argtypes = [C.c_void_p]
    L.ref_rank_create.restype = C.c_void_p; L.ref_rank_create.argtypes = [C.c_void_p, C.c_uint64]
    L.ref_rank_query.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.ref_rank_free.argtypes = [C.c_void_p]
    return L


def _ref():
    """the harness library for tests outside this module (None when it is neither built nor buildable here)"""
    if not os.path.exists(REF_SO):
        if not os.path.isdir("/root/reference/src/ksw2pp"):
            return None
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "-f", "Makefile.ref"])
    return C.CDLL(REF_SO)


def check_spp64(L, blob, expect):
    """`blob` (a serialised spp::sparse_hash_map<int64_t, int64_t>) loads in the reference's container and holds exactly `expect`"""
    import tempfile
    L.ref_spp64_load.restype = C.c_void_p; L.ref_spp64_load.argtypes = [C.c_char_p]
    L.ref_spp64_size.restype = C.c_int64; L.ref_spp64_size.argtypes = [C.c_void_p]
    L.ref_spp64_find.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    L.ref_spp64_free.argtypes = [C.c_void_p]
    with tempfile.NamedTemporaryFile(suffix=".spp") as f:
        f.write(blob); f.flush()
        h = L.ref_spp64_load(f.name.encode())
    assert h
    try:
        assert L.ref_spp64_size(h) == len(expect)
        keys = np.array(sorted(expect) + [max(expect) + 1, -5, 1 << 40], dtype=np.int64)
        found = np.zeros(keys.size, dtype=np.uint8); val = np.zeros(keys.size, dtype=np.int64)
        L.ref_spp64_find(h, keys.ctypes.data, keys.size, found.ctypes.data, val.ctypes.data)
        assert found[:-3].all() and not found[-3:].any()
        assert [int(v) for v in val[:-3]] == [expect[int(k)] for k in keys[:-3]]
    finally:
        L.ref_spp64_free(h)


@pytest.fixture(scope="module")
def ref_index(tmp_path_factory):
    d = tmp_path_factory.mktemp("ref_index_o")
    with tarfile.open(os.path.join(GOLD, "sample_data", "ref_index.tar.gz")) as t:
        t.extractall(d)
    return {"dense": str(d / "dense"), "perfect": str(d / "perfect")}


SCHEMES = [(2, -4, 4, 2, 15), (2, -4, 4, 2, 5), (1, -1, 1, 1, 15), (2, -6, 5, 3, 33), (2, -4, 4, 2, 0), (4, -4, 6, 2, 20),
           (2, -4, 4, 2, 34), (2, -4, 4, 2, 64), (1, -1, 1, 1, 97), (2, -4, 4, 2, 150), (2, -4, 4, 2, 1000), (2, -4, 4, 2, -1),
           (1, 0, 25, 25, 15)]        # last: --mimicStrictBT2's scoring


@pytest.mark.parametrize("scheme", SCHEMES)
def test_ksw2_extension_score_equals_the_references(ref, oracle_mod, scheme):
    """the oracle's byte-exact emulation of ksw_extz2_sse41 (kswExtz2) == ksw2pp::KSW2Aligner(EXTENSION) as getAlnScore calls it"""
    from test_ksw_variants import _cases
    a, b, q_, e_, w = scheme
    ol = oracle_mod._lib()
    ol.qo_ksw_extz2.restype = C.c_int
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    bad = []
    for q, t in _cases(777 + w, 600):
        want = ref.ref_ksw_extension(lut[q].tobytes(), len(q), lut[t].tobytes(), len(t), a, b, q_, e_, w)
        got = ol.qo_ksw_extz2(len(q), q.ctypes.data_as(C.c_void_p), len(t), t.ctypes.data_as(C.c_void_p), a, b, q_, e_, w)
        if got != want:
            bad.append((len(q), len(t), want, got))
    assert not bad, "%d of 600 differ, first %r" % (len(bad), bad[:5])


def test_ksw2_long_and_dirty_inputs(ref, oracle_mod):
    """256-base queries, lower case and IUPAC characters (seq_nt4_table maps them to 4), every target length residue mod 16"""
    ol = oracle_mod._lib()
    ol.qo_ksw_extz2.restype = C.c_int
    rng = np.random.default_rng(5)
    nt4 = np.full(256, 4, np.uint8)
    for i, c in enumerate(b"ACGT"):
        nt4[c] = i; nt4[c + 32] = i
    nt4[:4] = [0, 1, 2, 3]
    alphabet = np.frombuffer(b"ACGTacgtNnRYKM", dtype=np.uint8)
    for it in range(300):
        ql = int(rng.integers(200, 257)); tl = min(276, ql + 20 - int(rng.integers(0, 17)))
        q = alphabet[rng.choice(len(alphabet), ql, p=[.2, .2, .2, .2] + [.02] * 10)]
        t = q[:tl].copy() if tl <= ql else np.concatenate([q, alphabet[rng.integers(0, 4, tl - ql)]])
        m = rng.random(tl) < 0.04
        t[m] = alphabet[rng.integers(0, 4, int(m.sum()))]
        w = [15, 33, 60, -1][it % 4]
        want = ref.ref_ksw_extension(q.tobytes(), ql, t.tobytes(), tl, 2, -4, 4, 2, w)
        qc = nt4[q]; tc = nt4[t]
        got = ol.qo_ksw_extz2(ql, qc.ctypes.data_as(C.c_void_p), tl, tc.ctypes.data_as(C.c_void_p), 2, -4, 4, 2, w)
        assert got == want, (it, ql, tl, w, want, got)


def test_kmer_codec_equals_the_references(ref, oracle_mod):
    """fromChars (incl. its partial word and return value at the first non-ACGT character), reverse complement, homopolymer"""
    ol = oracle_mod._lib()
    ol.qo_kmer_rc.restype = C.c_uint64; ol.qo_kmer_rc.argtypes = [C.c_uint64, C.c_int]
    ol.qo_kmer_homopolymer.argtypes = [C.c_uint64, C.c_int]
    rng = np.random.default_rng(11)
    alphabet = np.frombuffer(b"ACGTacgtNnUuRY$", dtype=np.uint8)
    for k in (31, 21, 5, 1):
        for it in range(3000):
            p = [.22] * 4 + [.02] * 4 + [.01] * 7
            p = np.array(p) / sum(p)
            s = alphabet[rng.choice(len(alphabet), k, p=p)].tobytes() if it % 7 else bytes([b"ACGT"[it % 4]]) * k
            w, rc, hp = C.c_uint64(), C.c_uint64(), C.c_int()
            ok = ref.ref_kmer(s, k, C.byref(w), C.byref(rc), C.byref(hp))
            v = C.c_int()
            ow = ol.qo_kmer_encode(s, len(s), k, C.byref(v))
            assert v.value == ok and ow == w.value, (k, s, ok, v.value, hex(w.value), hex(ow))
            if ok:
                assert ol.qo_kmer_rc(ow, k) == rc.value, (k, s)
                assert ol.qo_kmer_homopolymer(ow, k) == hp.value, (k, s)


def test_boophf_lookup_equals_the_references(ref, ref_index):
    """the numpy BooPHF model the emulation and the device walk are checked against (oracle/q5ph.py) == boomphf::mphf::lookup on
    the .bph the reference wrote: every indexed k-mer and 20 000 k-mers that are not in the index"""
    from oracle import q5, q5ph
    path = os.path.join(ref_index["perfect"], "hash_info.bph")
    h = ref.ref_mphf_load(path.encode())
    assert h
    boo = q5ph.BooPHF(path)
    ix = q5.load(ref_index["dense"])
    keys = np.ascontiguousarray(ix.hkeys, dtype=np.uint64)
    rng = np.random.default_rng(3)
    other = rng.integers(0, 1 << 62, 20000, dtype=np.uint64)
    allk = np.concatenate([keys, other])
    out = np.zeros(allk.size, dtype=np.uint64)
    ref.ref_mphf_lookup(h, allk.ctypes.data, allk.size, out.ctypes.data)
    NOT_FOUND = (1 << 64) - 1                                # mphf::lookup returns ULLONG_MAX for a key the final hash does not hold
    mine = np.array([NOT_FOUND if (v := boo.lookup(int(x))) is None else v for x in allk], dtype=np.uint64)
    ref.ref_mphf_free(h)
    assert np.array_equal(out, mine)
    inidx = out[: keys.size]
    assert inidx.max() < keys.size and np.unique(inidx).size == keys.size          # a minimal perfect hash on its own keys


def test_rank_equals_the_references(ref, ref_index, oracle_mod):
    """rank9b::rank over rsd.bin (transcriptAtPosition, src/RapMapSAIndex.cpp:92-94) == the oracle's rank == the transcript whose
    text span holds the position (what the device's precomputed (tid, pos) table encodes)"""
    from oracle import oracle, q5
    raw = open(os.path.join(ref_index["dense"], "rsd.bin"), "rb").read()
    nbits = int(np.frombuffer(raw[:8], dtype="<u8")[0])
    bits = np.frombuffer(raw[8:], dtype=np.uint8).copy()
    h = ref.ref_rank_create(bits.ctypes.data, nbits)
    ix = q5.load(ref_index["dense"])
    orc = oracle.Oracle(ix)
    pos = np.concatenate([np.arange(0, min(nbits, 3000)), np.random.default_rng(1).integers(0, nbits, 20000)]).astype(np.uint64)
    out = np.zeros(pos.size, dtype=np.uint64)
    ref.ref_rank_query(h, pos.ctypes.data, pos.size, out.ctypes.data)
    ref.ref_rank_free(h)
    ol = oracle_mod._lib()
    mine = np.array([ol.qo_rank(orc.h, int(p)) for p in pos], dtype=np.uint64)
    assert np.array_equal(out, mine)
    offs = np.asarray(ix.txpOffsets, dtype=np.int64)
    text = np.asarray(ix.text)
    notsep = text[pos.astype(np.int64)] != ord("$")
    tid = np.searchsorted(offs, pos.astype(np.int64), side="right") - 1
    assert np.array_equal(out[notsep].astype(np.int64), tid[notsep])


def _spp(ref):
    ref.ref_spp_load.restype = C.c_void_p; ref.ref_spp_load.argtypes = [C.c_char_p]
    ref.ref_spp_size.restype = C.c_int64; ref.ref_spp_size.argtypes = [C.c_void_p]
    ref.ref_spp_dump.argtypes = [C.c_void_p] * 4
    ref.ref_spp_find.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    ref.ref_spp_free.argtypes = [C.c_void_p]
    ref.ref_xxh64.restype = C.c_uint64; ref.ref_xxh64.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
    return ref


def _spp_dump(ref, path):
    h = ref.ref_spp_load(path.encode())
    assert h, "the reference's sparse_hash_map could not unserialize " + path
    n = ref.ref_spp_size(h)
    keys = np.zeros(n, dtype=np.uint64); lb = np.zeros(n, dtype=np.int32); ub = np.zeros(n, dtype=np.int32)
    ref.ref_spp_dump(h, keys.ctypes.data, lb.ctypes.data, ub.ctypes.data)
    return h, keys, lb, ub


def test_dense_hash_file_equals_the_references_container(ref, ref_index, oracle_mod):
    """hash.bin written by the reference, read back by the reference's own spp::sparse_hash_map::unserialize
    (include/sparsepp/spp.h:2355-2429 through include/SparseHashSerializer.hpp:31-47, as RapMapSAIndex::load does,
    src/RapMapSAIndex.cpp:67-76): the oracle's numpy reader (oracle/q5.py) must see the same key -> interval map, and the
    oracle's find (what every parity test's lookups go through) must answer like the container's find, for present and absent keys"""
    from oracle import q5
    from conftest import load_oracle
    _spp(ref)
    d = ref_index["dense"]
    h, keys, lb, ub = _spp_dump(ref, os.path.join(d, "hash.bin"))
    try:
        k2, l2, u2 = q5.read_dense_hash(d)
        assert len(keys) == len(k2) == 18902
        o1 = np.argsort(keys); o2 = np.argsort(k2)
        assert np.array_equal(keys[o1], k2[o2]) and np.array_equal(lb[o1], l2[o2]) and np.array_equal(ub[o1], u2[o2])
        assert np.array_equal(keys, k2)                       # even the record order: the reader walks the file like the container
        ix, orc = load_oracle(d)
        rng = np.random.default_rng(3)
        probe = np.concatenate([keys, rng.integers(0, 1 << 62, 20000, dtype=np.uint64), keys ^ np.uint64(1)])
        found = np.zeros(len(probe), dtype=np.uint8); fl = np.zeros(len(probe), dtype=np.int32); fu = np.zeros(len(probe), dtype=np.int32)
        ref.ref_spp_find(h, probe.ctypes.data, len(probe), found.ctypes.data, fl.ctypes.data, fu.ctypes.data)
        assert found[: len(keys)].all() and not found[len(keys): len(keys) + 20000].any()
        ol = oracle_mod._lib()
        out = (C.c_int32 * 2)()
        for i in range(0, len(probe), 7):                    # every 7th probe through the oracle's own find
            got = ol.qo_hash_find(orc.h, int(probe[i]), out)
            assert bool(got) == bool(found[i])
            if got:
                assert (out[0], out[1]) == (fl[i], fu[i])
    finally:
        ref.ref_spp_free(h)


def test_our_hash_file_loads_in_the_references_container(ref, sample_data, lib_built):
    """hash.bin written by OUR index builder (rapmap_amd/csrc/qm_indexer.cpp) goes through the reference's unserialize, and the
    reference's find -- XXH64 + its own probing over the slots we chose -- returns every record: the file is a container the
    reference can use, not just the same map"""
    from oracle import q5
    _spp(ref)
    d = sample_data["idx"]
    h, keys, lb, ub = _spp_dump(ref, os.path.join(d, "hash.bin"))
    try:
        k2, l2, u2 = q5.read_dense_hash(d)
        assert np.array_equal(keys, k2) and np.array_equal(lb, l2) and np.array_equal(ub, u2) and len(keys) > 1000
        found = np.zeros(len(keys), dtype=np.uint8); fl = np.zeros(len(keys), dtype=np.int32); fu = np.zeros(len(keys), dtype=np.int32)
        ref.ref_spp_find(h, keys.ctypes.data, len(keys), found.ctypes.data, fl.ctypes.data, fu.ctypes.data)
        assert found.all() and np.array_equal(fl, lb) and np.array_equal(fu, ub)
    finally:
        ref.ref_spp_free(h)


def test_xxh64_equals_the_references(ref, lib_built):
    """the index builder's XXH64 (slot placement in hash.bin, duplicate filter) == src/xxhash.c, every length class of the
    algorithm (< 4, < 8, < 32, >= 32 bytes, unaligned tails), several seeds"""
    import rapmap_amd as ra
    _spp(ref)
    L = ra.api.lib()
    L.qm_xxh64.restype = C.c_uint64; L.qm_xxh64.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
    rng = np.random.default_rng(11)
    buf = rng.integers(0, 256, 4096, dtype=np.uint8)
    for ln in list(range(0, 100)) + [127, 128, 129, 1000, 4095]:
        for seed in (0, 1, 0x9E3779B97F4A7C15):
            for shift in (0, 1, 3):
                p = buf.ctypes.data + shift
                if shift + ln > buf.size:
                    continue
                assert L.qm_xxh64(p, ln, seed) == ref.ref_xxh64(p, ln, seed), (ln, seed, shift)
    keys = rng.integers(0, 1 << 62, 5000, dtype=np.uint64)
    for i in range(len(keys)):
        assert L.qm_xxh64(keys[i:].ctypes.data, 8, 0) == ref.ref_xxh64(keys[i:].ctypes.data, 8, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,compact", [("dense", False), ("perfect", False), ("perfect", True)])
def test_device_lookup_answers_like_the_references_container(ref, ref_index, lib_built, kind, compact):
    """the device's k-mer -> SA-interval lookup (bucket table built in HBM from hash.bin / from the -p files, or the BooPHF walk)
    against the REFERENCE's own container on the reference-written hash.bin: every indexed k-mer, given to the collector as
    a read of exactly k characters, must come back as the interval khash.find returns; random k-mers that the container does
    not hold (neither strand) must come back empty"""
    import rapmap_amd as ra
    _spp(ref)
    h, keys, lb, ub = _spp_dump(ref, os.path.join(ref_index["dense"], "hash.bin"))
    try:
        k = 31
        def decode(words):
            sh = np.arange(k - 1, -1, -1, dtype=np.uint64) * np.uint64(2)
            return np.frombuffer(b"ACGT", dtype=np.uint8)[((words[:, None] >> sh[None, :]) & np.uint64(3)).astype(np.int64)]
        def rc(words):
            c = decode(words)[:, ::-1]
            m = np.zeros(256, dtype=np.uint64); m[ord("A")] = 3; m[ord("C")] = 2; m[ord("G")] = 1; m[ord("T")] = 0
            sh = np.arange(k - 1, -1, -1, dtype=np.uint64) * np.uint64(2)
            return (m[c] << sh[None, :]).sum(axis=1).astype(np.uint64)
        rng = np.random.default_rng(5)
        absent = rng.integers(0, 1 << 62, 30000, dtype=np.uint64)
        f1 = np.zeros(len(absent), dtype=np.uint8); f2 = np.zeros(len(absent), dtype=np.uint8); t = np.zeros(len(absent), dtype=np.int32)
        ra_ = rc(absent)
        ref.ref_spp_find(h, absent.ctypes.data, len(absent), f1.ctypes.data, t.ctypes.data, t.ctypes.data)
        ref.ref_spp_find(h, ra_.ctypes.data, len(ra_), f2.ctypes.data, t.ctypes.data, t.ctypes.data)
        absent = absent[(f1 == 0) & (f2 == 0)]
        words = np.concatenate([keys, absent])
        seq = decode(words).reshape(-1).copy(); off = np.arange(len(words) + 1, dtype=np.int64) * k
        qi = ra.QuasiIndex(ref_index[kind]); mp = ra.QuasiMapper(qi, 0, ph_compact=compact)
        found, ioff, ints = mp.collect_reads(seq, off)
        cnt = np.diff(ioff)
        assert not cnt[len(keys):].any() and not found[len(keys):].any()
        checked = 0
        for i in range(len(keys)):
            if not (0 < ub[i] - lb[i] < 1000):
                continue                                    # the collector does not record intervals of >= maxInterval suffixes
            g = ints[ioff[i]:ioff[i + 1]]
            fw = g[(g["query_rc"] == 0) & (g["query_pos"] == 0)]
            if len(fw) == 0:                               # homopolymer k-mers are never probed (include/SACollector.hpp:176-181)
                w = int(keys[i]); assert w in (0, (1 << 62) - 1) or len(set(decode(keys[i:i + 1])[0].tolist())) == 1, hex(w)
                continue
            assert len(fw) == 1 and (int(fw["begin"][0]), int(fw["end"][0]), int(fw["len"][0])) == (int(lb[i]), int(ub[i]), k), i
            checked += 1
        assert checked > 18000
        mp.close(); qi.close()
    finally:
        ref.ref_spp_free(h)


def _ref_fastx(ref, p1, p2=None):
    ref.ref_fastx_dump.restype = C.c_int64; ref.ref_fastx_dump.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
    ref.ref_free.argtypes = [C.c_void_p]
    out = C.c_void_p()
    n = ref.ref_fastx_dump(p1.encode(), p2.encode() if p2 else None, C.byref(out))
    try:
        return [l.split(b"\t") for l in C.string_at(out, n).split(b"\n")[:-1]]
    finally:
        ref.ref_free(out)


def test_ingest_engine_reads_what_the_references_parser_reads(ref, lib_built, tmp_path):
    """the ingest engine (rapmap_amd/csrc/qm_ingest.cpp: chunk-parallel parse with record resync, gz blocks) against the
    REFERENCE's own fastx_parser::FastxParser<ReadPair / ReadSeq> (src/FastxParser.cpp over kseq, compiled in place) on the same
    files: the reference's sample reads (gz), a plain paired FASTQ cut into 3 kB chunks with qualities that start with '@'
    and '+', CRLF lines and ragged lengths, and a multi-line FASTA -- same records, same order, same sequences; our names
    are the whole header line, the reference's end at the first blank (where the SAM writer cuts ours, src/RapMapUtils.cpp:334-351)"""
    import random
    import gzip as gz
    import rapmap_amd as ra

    def ours(p1, p2, batch, threads):
        rd = ra.FastxReader(p1, p2, threads=threads)
        out = []
        for b in rd.chunks(batch):
            for i in range(b.n):
                rec = [bytes(b.names1[b.name_off1[i]:b.name_off1[i + 1]]).split(b" ")[0].split(b"\t")[0], bytes(b.seq1[b.off1[i]:b.off1[i + 1]])]
                if p2:
                    rec += [bytes(b.names2[b.name_off2[i]:b.name_off2[i + 1]]).split(b" ")[0].split(b"\t")[0], bytes(b.seq2[b.off2[i]:b.off2[i + 1]])]
                out.append(rec)
        rd.close()
        return out

    g1 = os.path.join(GOLD, "sample_data", "reads_1.fastq.gz"); g2 = os.path.join(GOLD, "sample_data", "reads_2.fastq.gz")
    want = _ref_fastx(ref, g1, g2)
    assert len(want) == 10000 and ours(g1, g2, 777, 3) == want
    assert ours(g1, None, 4096, 2) == _ref_fastx(ref, g1)
    rnd = random.Random(7)
    p1 = str(tmp_path / "a.fq"); p2 = str(tmp_path / "b.fq")
    with open(p1, "wb") as f1, open(p2, "wb") as f2:
        for i in range(20000):
            for f, nl in ((f1, b"\n"), (f2, b"\r\n" if i % 5 == 0 else b"\n")):
                L = rnd.choice([1, 30, 31, 100, 100, 250]) if i % 40 == 0 else 100
                s = "".join(rnd.choice("ACGTN") for _ in range(L)).encode()
                q = "".join(rnd.choice("@+I5#") for _ in range(L)).encode()
                nm = ("r%d desc text/%d" % (i * 7919, 1 + (f is f2))).encode()
                f.write(b"@" + nm + nl + s + nl + b"+" + nl + q + nl)
    os.environ["QM_INGEST_CHUNK"] = "3000"
    try:
        got = ours(p1, p2, 1000, 6)
    finally:
        del os.environ["QM_INGEST_CHUNK"]
    want = _ref_fastx(ref, p1, p2)
    assert len(want) == 20000 and got == want
    # the same pair as BGZF files (independent gzip members, inflated by helper threads side by side here; one multi-member gzip
    # stream to the reference's kseq / gzread)
    from util import write_bgzf
    b1 = str(tmp_path / "a.bgzf.fq.gz"); b2 = str(tmp_path / "b.bgzf.fq.gz")
    write_bgzf(b1, open(p1, "rb").read(), block=5000); write_bgzf(b2, open(p2, "rb").read(), block=65280)
    assert _ref_fastx(ref, b1, b2) == want and ours(b1, b2, 3000, 4) == want
    fa = str(tmp_path / "t.fa.gz")
    with gz.open(fa, "wt") as f:
        for i in range(3000):
            sq = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(1, 400)))
            f.write(">t%d gene=%d\n" % (i, i // 3))
            for j in range(0, len(sq), 60):
                f.write(sq[j:j + 60] + "\n")
    assert ours(fa, None, 257, 4) == _ref_fastx(ref, fa)
