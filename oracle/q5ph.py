"""oracle/q5ph.py -- TEST INFRASTRUCTURE: numpy model of RapMap's perfect-hash (`quasiindex -p`) files.

hash_info.bph = boomphf::mphf::save (include/BooPHF.hpp:1172-1197): f64 gamma, i32 nb_levels, u64 lastbitsetrank,
u64 nelem; per level bitVector::save (:773-781): u64 size, u64 nchar, nchar x u64, u64 nranks, nranks x u64;
then u64 final_n + final_n x (u64 key, u64 value).
hash_info.val = FrugalBooMap::save (include/FrugalBooMap.hpp:199-213): vector<IndexT> data_, vector<u8> lens_,
sparsepp-serialised overflow_ (IndexT -> IndexT).

lookup() below restates mphf::lookup / getLevel (BooPHF.hpp:971-1009,1318-1351) with hash64 (:394-407), the
xorshift `next` (:493-499), fastrange64 (:815-820) and bitVector::rank (:756-769); find() restates
FrugalBooMap::find (FrugalBooMap.hpp:149-167).
"""
import math
import os
import struct

import numpy as np

M64 = (1 << 64) - 1


def hash64(key, seed):
    h = seed
    h ^= ((h << 7) & M64) ^ ((key * (h >> 3)) & M64) ^ (~(((h << 11) + (key ^ (h >> 5))) & M64) & M64)
    h = ((~h & M64) + ((h << 21) & M64)) & M64
    h ^= h >> 24
    h = (h + ((h << 3) & M64) + ((h << 8) & M64)) & M64
    h ^= h >> 14
    h = (h + ((h << 2) & M64) + ((h << 4) & M64)) & M64
    h ^= h >> 28
    h = (h + ((h << 31) & M64)) & M64
    return h


class BooPHF:
    def __init__(self, path):
        b = open(path, "rb").read()
        off = 0
        self.gamma, = struct.unpack_from("<d", b, off); off += 8
        self.nb_levels, = struct.unpack_from("<i", b, off); off += 4
        self.lastbitsetrank, self.nelem = struct.unpack_from("<QQ", b, off); off += 16
        self.levels = []
        for _ in range(self.nb_levels):
            size, nchar = struct.unpack_from("<QQ", b, off); off += 16
            words = np.frombuffer(b, dtype="<u8", count=nchar, offset=off); off += 8 * nchar
            nr, = struct.unpack_from("<Q", b, off); off += 8
            ranks = np.frombuffer(b, dtype="<u8", count=nr, offset=off); off += 8 * nr
            self.levels.append((size, words, ranks))
        fn, = struct.unpack_from("<Q", b, off); off += 8
        self.final = {}
        for _ in range(fn):
            k, v = struct.unpack_from("<QQ", b, off); off += 16
            self.final[k] = v
        assert off == len(b), (off, len(b))
        # level domains are recomputed at load (BooPHF.hpp:1219-1230)
        n, g = float(self.nelem), self.gamma
        self.proba = 1.0 - math.pow((g * n - 1) / (g * n), self.nelem - 1)
        hd = int(math.ceil(n * g))
        self.domains = []
        for i in range(self.nb_levels):
            d = ((int(hd * math.pow(self.proba, i)) + 63) // 64) * 64
            self.domains.append(d if d else 64)

    def lookup(self, key):
        s = [0, 0]
        level = 0
        h = 0
        for ii in range(self.nb_levels - 1):
            if ii == 0:
                s[0] = hash64(key, 0xAAAAAAAA55555555); h = s[0]
            elif ii == 1:
                s[1] = hash64(key, 0x33333333CCCCCCCC); h = s[1]
            else:
                s1, s0 = s[0], s[1]
                s[0] = s0
                s1 ^= (s1 << 23) & M64
                s[1] = s1 ^ s0 ^ (s1 >> 17) ^ (s0 >> 26)
                h = (s[1] + s0) & M64
            pos = (h * self.domains[ii]) >> 64
            _, words, _ = self.levels[ii]
            if (int(words[pos >> 6]) >> (pos & 63)) & 1:
                break
            level += 1
        if level == self.nb_levels - 1:
            v = self.final.get(key)
            return None if v is None else v + self.lastbitsetrank
        _, words, ranks = self.levels[level]
        pos = (h * self.domains[level]) >> 64
        block = pos // 512
        r = int(ranks[block])
        for w in range(block * 8, pos >> 6):
            r += bin(int(words[w])).count("1")
        r += bin(int(words[pos >> 6]) & ((1 << (pos & 63)) - 1)).count("1")
        return r


def read_val(path, big=False):
    b = open(path, "rb").read()
    off = 0
    n, = struct.unpack_from("<Q", b, off); off += 8
    isz = 8 if big else 4
    data = np.frombuffer(b, dtype="<i8" if big else "<i4", count=n, offset=off).copy(); off += n * isz
    m, = struct.unpack_from("<Q", b, off); off += 8
    lens = np.frombuffer(b, dtype=np.uint8, count=m, offset=off).copy(); off += m

    def be():
        nonlocal off
        v = int.from_bytes(b[off:off + 4], "big"); off += 4
        if v == 0xFFFFFFFF:
            v = int.from_bytes(b[off:off + 8], "big"); off += 8
        return v
    magic = be(); ts = be(); nb = be()
    assert magic == 0x24687531
    off += ((ts + 31) // 32) * 4
    rec = np.frombuffer(b, dtype="<i8" if big else "<i4", count=2 * nb, offset=off).reshape(-1, 2)
    off += nb * 2 * isz
    assert off == len(b), (off, len(b))
    return data, lens, {int(a): int(c) for a, c in rec}


def enumerate_intervals(ix):
    """(keys, lb, ub) of a perfect-hash index = the run structure of the k-prefixes of the sorted suffixes
    (what buildPerfectHash feeds the MPHF, src/RapMapSAIndexer.cpp:129-215); FrugalBooMap::find is an exact map
    lookup, so the oracle can use the same table as for the dense index."""
    k = ix.k
    text, sa = ix.text, ix.SA.astype(np.int64)
    n = text.size
    code = np.full(256, 255, dtype=np.uint8)
    for i, c in enumerate(b"ACGT"):
        code[c] = i
    c2 = code[text]
    bad = (c2 == 255).astype(np.int64)
    cbad = np.concatenate([[0], np.cumsum(bad)])
    ok = (sa + k <= n)
    end = np.minimum(sa + k, n)
    ok &= (cbad[end] - cbad[sa]) == 0
    # 2-bit words of all text positions, then gather by SA.  Words of length 1, 2, 4, 8, 16 by doubling (w_2m[i] = w_m[i] << 2m | w_m[i + m]),
    # the k-mer word put together from the powers of two in k: a dozen passes over the text instead of k (3e8 characters: 8 s instead of 30)
    cc = np.zeros(n + k + 32, dtype=np.uint64)
    cc[:n] = c2.astype(np.uint64) & 3
    pw = {1: cc}
    m = 1
    while 2 * m <= k:
        a = pw[m]
        b = np.zeros_like(a)
        b[: a.size - m] = a[m:]
        b |= a << np.uint64(2 * m)
        pw[2 * m] = b
        m *= 2
    w = np.zeros(n + k + 32, dtype=np.uint64)
    done = 0
    for m in sorted(pw, reverse=True):
        if k - done >= m:
            w <<= np.uint64(2 * m)
            w[: w.size - done] |= pw[m][done:]
            done += m
    assert done == k
    del pw
    keys = w[sa]
    idx = np.nonzero(ok)[0]
    kk = keys[idx]
    brk = np.ones(idx.size, dtype=bool)
    brk[1:] = (kk[1:] != kk[:-1]) | (idx[1:] != idx[:-1] + 1)
    starts = np.nonzero(brk)[0]
    ends = np.concatenate([starts[1:], [idx.size]])
    return (np.ascontiguousarray(kk[starts]), idx[starts].astype(np.int32), (idx[ends - 1] + 1).astype(np.int32))
