"""Unit tests of the oracle's k-mer codec / reverseRead against direct definitions
(Kmer.hpp:40-51,92-100,484-487,525-542; RapMapUtils.cpp:63-72,107-128)."""
import ctypes as C

import numpy as np

CODE = {"A": 0, "C": 1, "G": 2, "T": 3}


def enc(s):
    w = 0
    for c in s:
        w = (w << 2) | CODE[c.upper()]
    return w


def test_encode_rc_homopolymer(oracle_mod):
    lib = oracle_mod._lib()
    rng = np.random.default_rng(0)
    for k in (31, 25, 15):
        for _ in range(200):
            s = "".join("ACGT"[i] for i in rng.integers(0, 4, k))
            v = C.c_int()
            w = lib.qo_kmer_encode(s.encode(), k, k, C.byref(v))
            assert v.value == 1 and w == enc(s)
            rc = s[::-1].translate(str.maketrans("ACGT", "TGCA"))
            assert lib.qo_kmer_rc(w, k) == enc(rc)
            assert lib.qo_kmer_homopolymer(w, k) == 0 or len(set(s)) == 1
        for b in "ACGT":
            assert lib.qo_kmer_homopolymer(enc(b * k), k) == 1
    # lower case is accepted, N / IUPAC stop the encoder leaving a partial word (high bits filled)
    v = C.c_int()
    assert lib.qo_kmer_encode(b"acgtacgtacgtacgtacgtacgtacgtacg", 31, 31, C.byref(v)) == enc("ACGTACGTACGTACGTACGTACGTACGTACG") and v.value == 1
    w = lib.qo_kmer_encode(b"ACGNACGTACGTACGTACGTACGTACGTACG", 31, 31, C.byref(v))
    assert v.value == 0 and w == enc("ACG") << (2 * 28)


def test_reverse_read(oracle_mod):
    import samfmt as sam
    lib = oracle_mod._lib()
    s = b"ACGTacgtNnUuRYKM-*xX"
    out = C.create_string_buffer(len(s))
    lib.qo_reverse_read(s, len(s), out)
    assert out.raw == b"NNNNNNNNAANNACGTACGT"
    assert sam.reverse_read(s) == out.raw


def test_two_bit_packing_round_trip(lib_built):
    """qm_pack_reads (the host-side packer the ingest engine's copy tasks share) against the numpy restatement of what the device's
    unpack kernels write: every character comes back -- upper-case A C G T out of the packed bytes, everything else (lower case,
    N, IUPAC codes, U, '$') out of the exception list -- for ragged lengths incl. empty reads, reads around the 32-character
    vector width and batches large enough for the threaded path; reads never share a packed byte"""
    import random
    import numpy as np
    from rapmap_amd import api
    rnd = random.Random(3)
    alphabet = "ACGT" * 12 + "acgtNnRYKMSWBDHVUu$"
    for nreads, lens in ((400, [0, 1, 2, 3, 4, 5, 31, 32, 33, 63, 64, 65, 100, 127, 250]), (70000, [100, 100, 100, 101, 36])):
        reads = []
        for i in range(nreads):
            L = rnd.choice(lens)
            dirty = rnd.random() < 0.3
            reads.append("".join(rnd.choice(alphabet if dirty else "ACGT") for _ in range(L)).encode())
        seq = np.frombuffer(b"".join(reads), dtype=np.uint8); off = np.zeros(nreads + 1, dtype=np.int64); off[1:] = np.cumsum([len(r) for r in reads])
        pk, o2, exc = api.pack_2bit(seq, off)
        assert np.array_equal(api.unpack_2bit(pk, off, exc), seq)
        want = np.nonzero(~np.isin(seq, np.frombuffer(b"ACGT", dtype=np.uint8)))[0]
        assert np.array_equal(np.sort(exc["pos"]), want.astype(np.uint32)) and np.array_equal(seq[exc["pos"]], exc["ch"].astype(np.uint8))
        start = (off[:-1] >> 2) + np.arange(nreads); end = start + (np.diff(off) + 3) // 4
        assert np.all(end[:-1] <= start[1:]) and end[-1] <= pk.size - 8
    # an exception list that is too small is an error the caller can act on (it then sends the characters)
    import ctypes as C
    seq = np.frombuffer(b"NNNNNNNN", dtype=np.uint8); off = np.array([0, 8], dtype=np.int64)
    pk = np.zeros(16, dtype=np.uint8); exc = np.zeros(4, dtype=api.PACK_EXC_DTYPE); ne = C.c_int64(0)
    assert api.lib().qm_pack_reads(seq.ctypes.data, off.ctypes.data, 1, pk.ctypes.data, exc.ctypes.data, 4, C.byref(ne)) != 0 and ne.value == 8
