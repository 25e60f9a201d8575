"""A transcriptome of the bench's KIND -- the GENCODE-like synthetic generator (rapmap_amd/synth.py: genes with shared exons, several
isoforms each), at a size the CPU suite can hold -- indexed by the product's own builder and cross-checked three ways, none of
which is the product checking itself (profiles/r04/big_index_crosscheck.py ran the same at config-2 size, 152 k transcripts, on
the GPU box's host; QMAP_BIGIDX_GENES sets the size here):
 (1) every array the product's loader (qm_index_open: mmap + views) holds against what the oracle's independent numpy reader
     (oracle/q5.py) parses out of the same files;
 (2) hash.bin through the REFERENCE's container -- spp::sparse_hash_map::unserialize compiled in place (oracle/_ref) -- whose
     find() must return every record with the same interval, and nothing for absent keys;
 (3) the suffix array against the text itself (sampled adjacent suffixes in order) and the hash intervals against the suffix
     array (first / last suffix carry the k-mer, the neighbours outside do not)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def big_index(tmp_path_factory):
    import rapmap_amd as ra
    from rapmap_amd import synth
    genes = int(os.environ.get("QMAP_BIGIDX_GENES", "6000"))
    d = tmp_path_factory.mktemp("bigidx")
    names, txps = synth.make_transcriptome(genes, seed=42)
    fa = str(d / "t.fa"); synth.write_fasta(fa, names, txps)
    idx = str(d / "idx")
    ra.build_index(fa, idx, k=31, threads=min(8, os.cpu_count() or 1))
    os.remove(fa)
    return idx


def test_loader_arrays_equal_the_independent_readers(big_index):
    import rapmap_amd as ra
    from oracle import q5
    ox = q5.load(big_index)
    qi = ra.QuasiIndex(big_index)
    try:
        text, offs = qi.arrays()
        assert qi.n_txps > 15000 and qi.text_len > 20_000_000
        assert np.array_equal(text, ox.text) and np.array_equal(offs, ox.txpOffsets.astype(np.int64))
        assert np.array_equal(qi.txp_lens, ox.txpLens) and qi.txp_names == list(ox.names)
        assert np.array_equal(qi.raw("complete_lens"), ox.completeLens)
        assert np.array_equal(qi.raw("sa"), ox.SA.astype(np.uint32))
        h = qi.raw("hash")
        assert np.array_equal(h["key"], ox.hkeys) and np.array_equal(h["lb"], ox.hlb.astype(np.uint32)) and np.array_equal(h["ub"], ox.hub.astype(np.uint32))
    finally:
        qi.close()


def test_suffix_array_and_hash_intervals_against_the_text(big_index):
    import rapmap_amd as ra
    qi = ra.QuasiIndex(big_index)
    try:
        text, _ = qi.arrays(); text = np.asarray(text); sa = np.asarray(qi.raw("sa")); h = qi.raw("hash")
        rng = np.random.default_rng(1)
        n = text.size
        i = rng.integers(0, sa.size - 1, size=100_000)
        W = 48

        def pref(pos):
            m = pos[:, None].astype(np.int64) + np.arange(W)[None, :]
            a = text[np.minimum(m, n - 1)].copy(); a[m >= n] = 0
            return a
        a = pref(sa[i]); b = pref(sa[i + 1])
        neq = a != b
        first = np.where(neq.any(1), neq.argmax(1), W)
        rows = np.arange(i.size)
        ordered = (first == W) | (a[rows, np.minimum(first, W - 1)] < b[rows, np.minimum(first, W - 1)])
        assert ordered.all()
        j = rng.integers(0, h.size, size=50_000)
        code = np.zeros(256, np.uint64); code[ord("C")] = 1; code[ord("G")] = 2; code[ord("T")] = 3
        valid = np.zeros(256, bool); valid[[ord(c) for c in "ACGT"]] = True

        def kmer_at(pos):
            m = pos[:, None].astype(np.int64) + np.arange(31)[None, :]
            inb = (m < n).all(1); m = np.minimum(m, n - 1)
            ch = text[m]; ok = inb & valid[ch].all(1)
            w = np.zeros(pos.size, np.uint64)
            for t_ in range(31):
                w = (w << np.uint64(2)) | code[ch[:, t_]]
            return w, ok
        lb = h["lb"][j].astype(np.int64); ub = h["ub"][j].astype(np.int64)
        w0, ok0 = kmer_at(sa[lb]); w1, ok1 = kmer_at(sa[ub - 1])
        assert (ok0 & ok1 & (w0 == h["key"][j]) & (w1 == h["key"][j])).all()
        wb, okb = kmer_at(sa[np.maximum(lb - 1, 0)]); wa, oka = kmer_at(sa[np.minimum(ub, sa.size - 1)])
        assert (((lb == 0) | ~okb | (wb != h["key"][j])) & ((ub == sa.size) | ~oka | (wa != h["key"][j]))).all()
    finally:
        qi.close()


def test_hash_file_loads_in_the_references_container(big_index):
    import rapmap_amd as ra
    refso = os.path.join(ROOT, "oracle", "_ref", "libqm_ref.so")
    if not os.path.exists(refso):
        pytest.skip("oracle/_ref/libqm_ref.so not built (the reference tree is not on this box)")
    qi = ra.QuasiIndex(big_index)
    try:
        h = qi.raw("hash")
        R = C.CDLL(refso)
        R.ref_spp_load.restype = C.c_void_p; R.ref_spp_load.argtypes = [C.c_char_p]
        R.ref_spp_size.restype = C.c_int64; R.ref_spp_size.argtypes = [C.c_void_p]
        R.ref_spp_find.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        R.ref_spp_free.argtypes = [C.c_void_p]
        hd = R.ref_spp_load(os.path.join(big_index, "hash.bin").encode())
        assert hd, "the reference's sparse_hash_map could not unserialize hash.bin"
        keys = np.ascontiguousarray(h["key"]); K = keys.size
        assert int(R.ref_spp_size(hd)) == K
        found = np.zeros(K, np.uint8); fl = np.zeros(K, np.int32); fu = np.zeros(K, np.int32)
        R.ref_spp_find(hd, keys.ctypes.data, K, found.ctypes.data, fl.ctypes.data, fu.ctypes.data)
        assert ((found == 1) & (fl.view(np.uint32) == h["lb"]) & (fu.view(np.uint32) == h["ub"])).all()
        rng = np.random.default_rng(2)
        absent = rng.integers(0, 1 << 62, size=200_000, dtype=np.uint64)
        absent = absent[~np.isin(absent, keys)]
        fa_ = np.zeros(absent.size, np.uint8); x = np.zeros(absent.size, np.int32); y = np.zeros(absent.size, np.int32)
        R.ref_spp_find(hd, absent.ctypes.data, absent.size, fa_.ctypes.data, x.ctypes.data, y.ctypes.data)
        R.ref_spp_free(hd)
        assert fa_.sum() == 0
    finally:
        qi.close()
