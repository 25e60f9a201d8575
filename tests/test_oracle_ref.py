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
