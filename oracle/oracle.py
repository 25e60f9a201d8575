"""oracle/oracle.py -- ctypes face of libqm_oracle.so (TEST INFRASTRUCTURE ONLY).

Usage:
    from oracle import q5, oracle
    ix = q5.load(index_dir)
    orc = oracle.Oracle(ix)
    res = orc.map_pairs(seq1, off1, seq2, off2, opts=oracle.default_opts(), nthreads=8)
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libqm_oracle.so")
_LIB64 = os.path.join(_HERE, "libqm_oracle64.so")     # IndexT = int64_t: the instantiation of a BigSA index

HIT_DTYPE = np.dtype([
    ("tid", "<u4"), ("pos", "<i4"), ("mate_pos", "<i4"), ("frag_len", "<u4"),
    ("read_len", "<u4"), ("mate_len", "<u4"),
    ("fwd", "u1"), ("mate_is_fwd", "u1"), ("is_paired", "u1"), ("mate_status", "u1"),
    ("aln_score", "<i4"),
])
assert HIT_DTYPE.itemsize == 32


class Opts(C.Structure):
    _fields_ = [("sensitive", C.c_int32), ("strictCheck", C.c_int32), ("maxNumHits", C.c_int32),
                ("noOrphans", C.c_int32), ("noDovetail", C.c_int32), ("fuzzy", C.c_int32),
                ("maxInterval", C.c_int32), ("pad", C.c_int32), ("quasiCov", C.c_double),
                ("selAln", C.c_int32), ("hardFilter", C.c_int32), ("matchScore", C.c_int32), ("mismatchPenalty", C.c_int32),
                ("gapOpen", C.c_int32), ("gapExtend", C.c_int32), ("dpBandwidth", C.c_int32), ("maxMMPExtension", C.c_int32),
                ("alnPolicy", C.c_int32), ("recoverOrphans", C.c_int32),
                ("minScoreFraction", C.c_double), ("consensusSlack", C.c_double)]


def default_opts(**kw):
    """Defaults of `rapmap quasimap` (src/RapMapSAMapper.cpp:992-1023,1113-1114)."""
    o = Opts(sensitive=1, strictCheck=1, maxNumHits=200, noOrphans=0, noDovetail=0, fuzzy=0,
             maxInterval=1000, pad=0, quasiCov=0.0,
             # -s sub-options (src/RapMapSAMapper.cpp:1011-1023); only read when selAln is set
             selAln=0, hardFilter=0, matchScore=2, mismatchPenalty=-4, gapOpen=4, gapExtend=2, dpBandwidth=15,
             maxMMPExtension=7, alnPolicy=0, recoverOrphans=0, minScoreFraction=0.65, consensusSlack=0.2)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def mimic_bt2_opts(strict=False, **kw):
    """--mimicBT2 / --mimicStrictBT2 (src/RapMapSAMapper.cpp:1149-1173)"""
    o = default_opts(selAln=1, alnPolicy=2 if strict else 1, noOrphans=1, noDovetail=1, consensusSlack=0.35, maxNumHits=1000)
    if strict:
        o.minScoreFraction = 0.8; o.matchScore = 1; o.mismatchPenalty = 0; o.gapOpen = 25; o.gapExtend = 25
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _lib(big=False):
    path = _LIB64 if big else _LIB
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(_HERE, "qm_oracle.cpp")):
        build()
    lib = C.CDLL(path)
    lib.qo_index_bytes.restype = C.c_int
    assert lib.qo_index_bytes() == (8 if big else 4)
    lib.qo_index_create.restype = C.c_void_p
    lib.qo_index_create.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                    C.c_int64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_int64]
    lib.qo_index_destroy.argtypes = [C.c_void_p]
    lib.qo_map.restype = C.c_int
    lib.qo_map.argtypes = [C.c_void_p, C.POINTER(Opts), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                           C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p,
                           C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.qo_free.argtypes = [C.c_void_p]
    lib.qo_kmer_encode.restype = C.c_uint64
    lib.qo_kmer_encode.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.POINTER(C.c_int)]
    lib.qo_kmer_rc.restype = C.c_uint64
    lib.qo_kmer_rc.argtypes = [C.c_uint64, C.c_int]
    lib.qo_kmer_homopolymer.restype = C.c_int
    lib.qo_kmer_homopolymer.argtypes = [C.c_uint64, C.c_int]
    lib.qo_reverse_read.argtypes = [C.c_char_p, C.c_int64, C.c_char_p]
    lib.qo_extend_search.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_char_p, C.c_int64,
                                     C.c_void_p]
    lib.qo_hash_find.restype = C.c_int
    lib.qo_hash_find.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    lib.qo_kpos_ties.argtypes = [C.POINTER(C.c_longlong), C.c_int]
    lib.qo_kpos_ties.restype = None
    lib.qo_rank.restype = C.c_uint64
    lib.qo_rank.argtypes = [C.c_void_p, C.c_uint64]
    return lib


def pack_reads(reads):
    """list of bytes/str -> (uint8 concat, int64 offsets[n+1])"""
    bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
    off = np.zeros(len(bs) + 1, dtype=np.int64)
    np.cumsum([len(b) for b in bs], out=off[1:])
    seq = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, np.uint8)
    return seq, off


class MapResult:
    pass


class Oracle:
    def __init__(self, ix):
        self.lib = _lib(bool(ix.big))
        self.ix = ix
        it = np.int64 if ix.big else np.int32              # RapMapSAIndex<IndexT, ...>::IndexType
        # keep references alive
        self._text = np.ascontiguousarray(ix.text, dtype=np.uint8)
        self._sa = np.ascontiguousarray(ix.SA, dtype=it)
        self._off = np.ascontiguousarray(ix.txpOffsets, dtype=it)
        self._rsd = np.ascontiguousarray(ix.rsd, dtype=np.uint64)
        self._hk = np.ascontiguousarray(ix.hkeys, dtype=np.uint64)
        self._hlb = np.ascontiguousarray(ix.hlb, dtype=it)
        self._hub = np.ascontiguousarray(ix.hub, dtype=it)
        self.h = self.lib.qo_index_create(
            ix.k, self._text.ctypes.data, self._text.size, self._sa.ctypes.data, self._sa.size,
            self._off.ctypes.data, self._off.size, self._rsd.ctypes.data, ix.nbits,
            self._hk.ctypes.data, self._hlb.ctypes.data, self._hub.ctypes.data, self._hk.size)

    def __del__(self):
        try:
            self.lib.qo_index_destroy(self.h)
        except Exception:
            pass

    def _map(self, seq1, off1, seq2, off2, opts, nthreads, want_ints):
        opts = opts or default_opts()
        n = len(off1) - 1
        seq1 = np.ascontiguousarray(seq1, dtype=np.uint8)
        off1 = np.ascontiguousarray(off1, dtype=np.int64)
        paired = seq2 is not None
        if paired:
            seq2 = np.ascontiguousarray(seq2, dtype=np.uint8)
            off2 = np.ascontiguousarray(off2, dtype=np.int64)
            assert len(off2) - 1 == n
        hit_off = np.zeros(n + 1, dtype=np.int64)
        counters = np.zeros(6, dtype=np.uint64)
        work = np.zeros(10, dtype=np.uint64)
        hits_p = C.c_void_p()
        ints_off = np.zeros(n + 1, dtype=np.int64)
        ints_p = C.c_void_p()
        rc = self.lib.qo_map(self.h, C.byref(opts), n, seq1.ctypes.data, off1.ctypes.data,
                             seq2.ctypes.data if paired else None, off2.ctypes.data if paired else None,
                             nthreads, hit_off.ctypes.data, C.byref(hits_p), counters.ctypes.data,
                             work.ctypes.data, ints_off.ctypes.data if want_ints else None,
                             C.byref(ints_p) if want_ints else None)
        assert rc == 0
        total = int(hit_off[-1])
        hits = np.ctypeslib.as_array(C.cast(hits_p, C.POINTER(C.c_uint8)), shape=(max(total, 1) * 32,))
        hits = hits[: total * 32].copy().view(HIT_DTYPE)
        self.lib.qo_free(hits_p)
        r_map_s, r_call_s = float(work[8]) * 1e-9, float(work[9]) * 1e-9
        r = MapResult()
        r.hit_offsets, r.hits = hit_off, hits
        r.counters = dict(zip(["peHits", "seHits", "totHits", "numReads", "tooManyHits", "mappedUnits"],
                              [int(x) for x in counters]))
        r.work = dict(zip(["n_probe", "n_sa", "n_text", "n_rank", "n_hits", "n_aln", "n_cells", "n_ungapped"], [int(x) for x in work[:8]]))
        # seconds of the mapping section alone (worker threads started .. joined: the span the reference's own timer covers)
        # and of the whole native call (+ one contiguous result array)
        r.map_seconds, r.call_seconds = r_map_s, r_call_s
        if want_ints:
            tot = int(ints_off[-1])
            a = np.ctypeslib.as_array(C.cast(ints_p, C.POINTER(C.c_int32)), shape=(max(tot, 1) * 6,))
            r.ints = a[: tot * 6].copy().reshape(-1, 6)
            r.ints_offsets = ints_off
            self.lib.qo_free(ints_p)
        return r

    def map_pairs(self, seq1, off1, seq2, off2, opts=None, nthreads=1, want_ints=False):
        return self._map(seq1, off1, seq2, off2, opts, nthreads, want_ints)

    def map_single(self, seq, off, opts=None, nthreads=1):
        return self._map(seq, off, None, None, opts, nthreads, False)


def kpos_ties(reset=True, big=False):
    """the --noSensitive vote (SACollector.hpp:289-337) since the last reset: dict(ties = positions that entered kmerScores more
    than once, conflicts = ties whose entries carry different scores, conflicts_without_U = ... in a window without U / u,
    conflict_reads, decision_differs = reads where libstdc++'s unstable std::sort -- what the reference calls -- would have
    decided the strand differently from the stable sort of the restatement)"""
    a = (C.c_longlong * 5)()
    _lib(big).qo_kpos_ties(a, 1 if reset else 0)
    return dict(zip(("ties", "conflicts", "conflicts_without_U", "conflict_reads", "decision_differs"), [int(x) for x in a]))
