"""oracle/q5.py -- TEST INFRASTRUCTURE (numpy reader of RapMap's on-disk quasi-index "q5").

Independent of the product's C++ loader (rapmap_amd/csrc/qm_index.cpp) on purpose:
the oracle and the HIP path read the same bytes through two different parsers, so
a mis-parse on either side shows up as a parity failure.

Byte layouts follow SURVEY.md Appendix A, i.e. the writers in
  src/RapMapSAIndexer.cpp:109-110,242-243 (sa.bin), :694-731 (rsd.bin, txpInfo.bin),
  :433-441 + include/sparsepp/spp.h:2355-2366,2420-2429 (hash.bin),
  include/IndexHeader.hpp:45-57 (header.json).
cereal binary archives store a vector<arith>/string as u64 count + raw bytes.
"""
import json
import os

import numpy as np


class Q5Index:
    pass


def _u64(buf, off):
    return int(np.frombuffer(buf, dtype="<u8", count=1, offset=off)[0]), off + 8


def read_header(d):
    with open(os.path.join(d, "header.json")) as f:
        h = json.load(f)["value0"]
    return h


def read_sa(d, big=False):
    p = os.path.join(d, "sa.bin")
    n = int(np.fromfile(p, dtype="<u8", count=1)[0])
    dt = "<i8" if big else "<i4"
    sa = np.fromfile(p, dtype=dt, count=n, offset=8)
    assert sa.size == n
    return sa


def read_txpinfo(d, big=False):
    buf = np.fromfile(os.path.join(d, "txpInfo.bin"), dtype=np.uint8)
    off = 0
    cnt, off = _u64(buf, off)
    names = []
    for _ in range(cnt):
        ln, off = _u64(buf, off)
        names.append(bytes(buf[off:off + ln]).decode())
        off += ln
    cnt2, off = _u64(buf, off)
    isz = 8 if big else 4
    offsets = np.frombuffer(buf, dtype="<i8" if big else "<i4", count=cnt2, offset=off).copy()
    off += cnt2 * isz
    tl, off = _u64(buf, off)
    text = buf[off:off + tl].copy()
    off += tl
    cnt3, off = _u64(buf, off)
    complete = np.frombuffer(buf, dtype="<u4", count=cnt3, offset=off).copy()
    off += cnt3 * 4
    assert off == buf.size, (off, buf.size)
    return names, offsets, text, complete


def read_rsd(d):
    p = os.path.join(d, "rsd.bin")
    nbits = int(np.fromfile(p, dtype="<u8", count=1)[0])
    nbytes = (nbits + 7) // 8
    raw = np.fromfile(p, dtype=np.uint8, count=nbytes, offset=8)
    nwords = (nbits + 63) // 64
    padded = np.zeros(nwords * 8 + 8, dtype=np.uint8)   # +1 spare word
    padded[:nbytes] = raw
    return nbits, padded.view("<u8")


def read_dense_hash(d, big=False):
    """-> (keys u64[K], lb[K], ub[K]) in table order."""
    p = os.path.join(d, "hash.bin")
    with open(p, "rb") as f:
        def be32or64():
            v = int.from_bytes(f.read(4), "big")
            if v == 0xFFFFFFFF:
                v = int.from_bytes(f.read(8), "big")
            return v
        magic = be32or64()
        assert magic == 0x24687531, hex(magic)
        table_size = be32or64()
        num_buckets = be32or64()
        ngroups = (table_size + 31) // 32
        f.seek(ngroups * 4, 1)
        start = f.tell()
    isz = 8 if big else 4
    rec = np.dtype([("key", "<u8"), ("lb", "<i8" if big else "<i4"), ("ub", "<i8" if big else "<i4")])
    assert rec.itemsize == 8 + 2 * isz
    recs = np.fromfile(p, dtype=rec, count=num_buckets, offset=start)
    assert recs.size == num_buckets
    assert os.path.getsize(p) == start + num_buckets * rec.itemsize
    return (np.ascontiguousarray(recs["key"]), np.ascontiguousarray(recs["lb"]),
            np.ascontiguousarray(recs["ub"]))


def load(d, enum_cache=None):
    """enum_cache (a path prefix, -p indices only): where the enumerated k-mer table of the index is kept between processes -- bench.py's
    background child writes it beside the index it builds, the bench process reads it instead of spending 20 s of numpy on it"""
    d = d.rstrip("/") + "/"
    h = read_header(d)
    ix = Q5Index()
    ix.dir = d
    ix.header = h
    ix.k = int(h["KmerLen"])
    ix.big = bool(h["BigSA"])
    ix.perfect = bool(h["PerfectHash"])
    ix.SA = read_sa(d, ix.big)
    ix.names, ix.txpOffsets, ix.text, ix.completeLens = read_txpinfo(d, ix.big)
    ix.nbits, ix.rsd = read_rsd(d)
    # src/RapMapSAIndex.cpp:151-163
    lens = np.empty(len(ix.txpOffsets), dtype=np.int64)
    lens[:-1] = ix.txpOffsets[1:].astype(np.int64) - 1 - ix.txpOffsets[:-1]
    lens[-1] = ix.SA.size - 1 - int(ix.txpOffsets[-1])
    ix.txpLens = lens
    if not ix.perfect:
        ix.hkeys, ix.hlb, ix.hub = read_dense_hash(d, ix.big)
    else:
        from oracle import q5ph  # noqa
        import os
        names = [enum_cache + "_%s.npy" % t for t in ("keys", "lb", "ub")] if enum_cache else []
        if names and all(os.path.exists(f) for f in names):
            ix.hkeys, ix.hlb, ix.hub = (np.load(f) for f in names)
        else:
            ix.hkeys, ix.hlb, ix.hub = q5ph.enumerate_intervals(ix)
            if names:
                for f, a in zip(names, (ix.hkeys, ix.hlb, ix.hub)):
                    with open(f + ".tmp", "wb") as fh:      # (whoever finds the file finds all of it)
                        np.save(fh, a)
                    os.replace(f + ".tmp", f)
    return ix
