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
