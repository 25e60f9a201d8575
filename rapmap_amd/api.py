"""Host-side mirror of the reference's call surface, over the C ABI of libqmap_mi355.so.

The reference exposes three C++ entry points to its callers:
``SACollector::operator()`` (include/SACollector.hpp:108), ``hitsToMappingsSimple``
(include/HitManager.hpp:130-135) and ``mergeLeftRightHits`` (include/RapMapUtils.hpp:1185),
driven per read pair by ``processReadsPairSA`` (src/RapMapSAMapper.cpp:376).  Here the same
three steps run batched on the GPU behind ``QuasiMapper.map_pairs``; the index object mirrors
``RapMapSAIndex`` (include/RapMapSAIndex.hpp:70-82).

There is NO CPU fallback: if the HIP library is missing or no GPU is visible the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QM_LIB_OVERRIDE") or os.path.join(_HERE, "libqmap_mi355.so")   # override: profiling variants only

HIT_DTYPE = np.dtype([
    ("tid", "<u4"), ("pos", "<i4"), ("mate_pos", "<i4"), ("frag_len", "<u4"),
    ("read_len", "<u4"), ("mate_len", "<u4"),
    ("fwd", "u1"), ("mate_is_fwd", "u1"), ("is_paired", "u1"), ("mate_status", "u1"),
    ("aln_score", "<i4"),
])
INTERVAL_DTYPE = np.dtype([("begin", "<i4"), ("end", "<i4"), ("len", "<u4"), ("query_pos", "<u4"),
                           ("query_rc", "u1"), ("list", "u1"), ("pad", "<u2")])
assert HIT_DTYPE.itemsize == 32 and INTERVAL_DTYPE.itemsize == 20

# every symbol include/qmap_mi355.h declares
ABI_SYMBOLS = [
    "qm_last_error", "qm_version", "qm_opts_default", "qm_index_open", "qm_index_close", "qm_index_info_get",
    "qm_index_txp_name", "qm_index_txp_len", "qm_index_arrays", "qm_index_raw", "qm_ctx_create", "qm_ctx_destroy", "qm_ctx_device_bytes",
    "qm_map_pairs", "qm_map_reads", "qm_map_device", "qm_pack_reads", "qm_packed_offset", "qm_packed_bytes", "qm_map_pairs_packed", "qm_map_reads_packed", "qm_fetch_hits", "qm_result_device", "qm_ctx_set_debug",
    "qm_fetch_intervals", "qm_last_kernel_ms", "qm_ctx_stat", "qm_fetch_skipped", "qm_build_index", "qm_build_index_ex",
    "qm_collect_reads", "qm_fetch_found", "qm_hits_to_mappings", "qm_fetch_read_lists", "qm_merge_lists", "qm_fetch_too_many",
    "qm_map_pairs_stages", "qm_map_pairs_stages_ex", "qm_map_pairs_stages_packed", "qm_stage_bytes", "qm_fetch_stages", "qm_pinned_alloc", "qm_pinned_free", "qm_ctx_create_ex", "qm_fetch_hits_pinned", "qm_xxh64",
    "qm_stream_open", "qm_stream_open_ex", "qm_stream_reserve", "qm_stream_next", "qm_stream_close", "qm_stream_last_error", "qm_stream_stats", "qm_stream_stats_ex",
    "qm_reader_open", "qm_reader_next", "qm_reader_close", "qm_io_last_error", "qm_sam_header", "qm_sam_records",
    "qm_sam_write", "qm_sam_writer_open", "qm_sam_writer_open_ex", "qm_sam_writer_header", "qm_sam_writer_put", "qm_sam_writer_close", "qm_buf_free",
]


class QmError(RuntimeError):
    pass


class QmOpts(C.Structure):
    """MappingOpts (src/RapMapSAMapper.cpp:114-152), hot-path fields only."""
    _fields_ = [("sensitive", C.c_int32), ("strict_check", C.c_int32), ("max_num_hits", C.c_int32),
                ("no_orphans", C.c_int32), ("no_dovetail", C.c_int32), ("fuzzy", C.c_int32),
                ("max_interval", C.c_int32), ("sel_aln", C.c_int32), ("quasi_cov", C.c_double),
                ("hard_filter", C.c_int32), ("match_score", C.c_int32), ("mismatch_penalty", C.c_int32), ("gap_open", C.c_int32),
                ("gap_extend", C.c_int32), ("dp_bandwidth", C.c_int32), ("max_mmp_extension", C.c_int32), ("aln_policy", C.c_int32),
                ("min_score_fraction", C.c_double), ("consensus_slack", C.c_double)]


class QmCounters(C.Structure):
    """HitCounters (include/RapMapUtils.hpp:208-216)."""
    _fields_ = [("pe_hits", C.c_uint64), ("se_hits", C.c_uint64), ("tot_hits", C.c_uint64),
                ("num_reads", C.c_uint64), ("too_many_hits", C.c_uint64), ("mapped", C.c_uint64)]

    def as_dict(self):
        return {"peHits": self.pe_hits, "seHits": self.se_hits, "totHits": self.tot_hits,
                "numReads": self.num_reads, "tooManyHits": self.too_many_hits, "mappedUnits": self.mapped}


class QmHitRaw(C.Structure):
    _fields_ = [("b", C.c_uint8 * 32)]


class QmStreamBatch(C.Structure):
    """qm_stream_batch (include/qmap_mi355.h): pointers into the stream's pinned buffers"""
    _fields_ = [("n_units", C.c_int64),
                ("seq1", C.c_void_p), ("off1", C.c_void_p), ("names1", C.c_void_p), ("name_off1", C.c_void_p),
                ("seq2", C.c_void_p), ("off2", C.c_void_p), ("names2", C.c_void_p), ("name_off2", C.c_void_p),
                ("hit_offsets", C.c_void_p), ("hits", C.c_void_p), ("n_hits", C.c_int64),
                ("counters", QmCounters), ("gpu_ms", C.c_double), ("device", C.c_int32), ("pad", C.c_int32)]


class QmStageView(C.Structure):
    _fields_ = [("n_units", C.c_int64), ("n_reads", C.c_int64), ("iv_off", C.c_void_p), ("iv", C.c_void_p), ("found", C.c_void_p),
                ("list_off", C.c_void_p), ("words", C.c_void_p), ("hit_off", C.c_void_p), ("hits", C.c_void_p), ("too_many", C.c_void_p)]


class QmIndexInfo(C.Structure):
    _fields_ = [("k", C.c_int32), ("big_sa", C.c_int32), ("perfect_hash", C.c_int32), ("pad", C.c_int32),
                ("text_len", C.c_int64), ("n_txps", C.c_int64), ("n_keys", C.c_int64)]


_lib = None


def lib():
    """Load libqmap_mi355.so; fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise QmError("HIP library %s is missing - run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(there is no CPU fallback)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.qm_last_error.restype = C.c_char_p
    L.qm_version.restype = C.c_char_p
    L.qm_index_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
    L.qm_index_close.argtypes = [C.c_void_p]
    L.qm_index_info_get.argtypes = [C.c_void_p, C.POINTER(QmIndexInfo)]
    L.qm_index_txp_name.restype = C.c_char_p
    L.qm_index_txp_name.argtypes = [C.c_void_p, C.c_int64]
    L.qm_index_txp_len.restype = C.c_int64
    L.qm_index_txp_len.argtypes = [C.c_void_p, C.c_int64]
    L.qm_index_arrays.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_void_p),
                                  C.POINTER(C.c_int64)]
    L.qm_ctx_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    L.qm_ctx_create_ex.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.POINTER(C.c_void_p)]
    L.qm_ctx_destroy.argtypes = [C.c_void_p]
    L.qm_ctx_device_bytes.restype = C.c_int64
    L.qm_ctx_device_bytes.argtypes = [C.c_void_p]
    L.qm_map_pairs.argtypes = [C.c_void_p, C.POINTER(QmOpts), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.POINTER(C.c_int64), C.POINTER(QmCounters)]
    L.qm_map_reads.argtypes = [C.c_void_p, C.POINTER(QmOpts), C.c_int64, C.c_void_p, C.c_void_p,
                               C.POINTER(C.c_int64), C.POINTER(QmCounters)]
    L.qm_map_device.argtypes = [C.c_void_p, C.POINTER(QmOpts), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(QmCounters)]
    L.qm_fetch_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.qm_result_device.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.qm_ctx_set_debug.argtypes = [C.c_void_p, C.c_int]
    L.qm_fetch_intervals.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    L.qm_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.qm_ctx_stat.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]
    L.qm_fetch_skipped.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    L.qm_collect_reads.argtypes = [C.c_void_p, C.POINTER(QmOpts), C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
    L.qm_fetch_found.argtypes = [C.c_void_p, C.c_void_p]
    L.qm_hits_to_mappings.argtypes = [C.c_void_p, C.POINTER(QmOpts), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
    L.qm_fetch_read_lists.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    L.qm_merge_lists.argtypes = [C.c_void_p, C.POINTER(QmOpts), C.c_int64] + [C.c_void_p] * 8 + [C.POINTER(C.c_int64), C.POINTER(QmCounters)]
    L.qm_fetch_too_many.argtypes = [C.c_void_p, C.c_void_p]
    L.qm_map_pairs_stages.argtypes = L.qm_map_pairs.argtypes
    L.qm_pack_reads.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    L.qm_map_pairs_packed.argtypes = [C.c_void_p, C.POINTER(QmOpts), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(QmCounters)]
    L.qm_map_reads_packed.argtypes = [C.c_void_p, C.POINTER(QmOpts), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                      C.POINTER(C.c_int64), C.POINTER(QmCounters)]
    L.qm_stage_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    L.qm_fetch_stages.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(QmStageView)]
    L.qm_pinned_alloc.argtypes = [C.c_int64]; L.qm_pinned_alloc.restype = C.c_void_p
    L.qm_pinned_free.argtypes = [C.c_void_p]; L.qm_pinned_free.restype = None
    L.qm_stream_open.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.POINTER(QmOpts), C.c_char_p, C.c_char_p, C.c_int64, C.c_int32,
                                 C.POINTER(C.c_void_p)]
    L.qm_stream_open_ex.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int32, C.c_uint32, C.POINTER(QmOpts), C.c_char_p, C.c_char_p,
                                    C.c_int64, C.c_int32, C.c_uint32, C.POINTER(C.c_void_p)]
    L.qm_stream_reserve.argtypes = [C.c_int64]
    L.qm_stream_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.qm_stream_stats_ex.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int32]
    L.qm_stream_next.argtypes = [C.c_void_p, C.POINTER(QmStreamBatch)]
    L.qm_stream_close.argtypes = [C.c_void_p]
    L.qm_stream_last_error.restype = C.c_char_p
    L.qm_fetch_hits_pinned.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.qm_build_index.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    L.qm_build_index_ex.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_char_p]
    L.qm_opts_default.argtypes = [C.POINTER(QmOpts)]
    L.qm_io_last_error.restype = C.c_char_p
    L.qm_reader_open.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.POINTER(C.c_void_p)]
    L.qm_reader_next.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int64)] + [C.POINTER(C.c_void_p)] * 8
    L.qm_reader_close.argtypes = [C.c_void_p]
    L.qm_sam_header.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    L.qm_sam_records.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 10 + [C.c_int32, C.c_int32,
                                 C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    L.qm_sam_write.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 10 + [C.c_int32, C.c_int32, C.c_int,
                               C.POINTER(C.c_int64)]
    L.qm_sam_writer_open.argtypes = [C.c_void_p, C.c_int, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    L.qm_sam_writer_open_ex.argtypes = [C.c_void_p, C.c_int, C.c_int32, C.c_int32, C.c_uint32, C.POINTER(C.c_void_p)]
    L.qm_sam_writer_header.argtypes = [C.c_void_p]
    L.qm_sam_writer_put.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 10
    L.qm_sam_writer_close.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    L.qm_buf_free.argtypes = [C.c_void_p]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise QmError("qmap_mi355 error %d: %s" % (rc, lib().qm_last_error().decode(errors="replace")))


def default_opts(**kw):
    """`rapmap quasimap` defaults (src/RapMapSAMapper.cpp:992-1023,1113-1114)."""
    o = QmOpts()
    _check(lib().qm_opts_default(C.byref(o)))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def build_index(fasta, out_dir, k=31, no_clip_poly_a=False, keep_duplicates=False, threads=None, perfect_hash=False, header_sep=None):
    """`rapmap quasiindex -t FASTA -i OUT -k K [-p] [-s SEP]` (src/RapMapSAIndexer.cpp:821-927); int64 ("BigSA") form for a text beyond 2^31 - 2 characters."""
    threads = threads or min(32, os.cpu_count() or 1)
    rc = lib().qm_build_index_ex(os.fsencode(fasta), os.fsencode(out_dir), k, int(no_clip_poly_a), int(keep_duplicates), threads,
                                 int(perfect_hash), header_sep.encode() if header_sep is not None else None)
    if rc != 0:
        lib().qm_indexer_last_error.restype = C.c_char_p
        raise QmError("qm_build_index failed (%d): %s" % (rc, lib().qm_indexer_last_error().decode()))


def pack_reads(reads):
    """list of bytes/str -> (uint8 concat, int64 offsets[n+1])"""
    bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
    off = np.zeros(len(bs) + 1, dtype=np.int64)
    if bs:
        np.cumsum([len(b) for b in bs], out=off[1:])
    seq = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, np.uint8)
    return seq, off


class QuasiIndex:
    """RapMapSAIndex<int32_t | int64_t, RegHashT | PerfectHashT>: load() == qm_index_open (src/RapMapSAIndex.cpp:97-176)."""

    def __init__(self, index_dir):
        self._h = C.c_void_p()
        _check(lib().qm_index_open(os.fsencode(index_dir), C.byref(self._h)))
        info = QmIndexInfo()
        _check(lib().qm_index_info_get(self._h, C.byref(info)))
        self.k, self.text_len, self.n_txps, self.n_keys = info.k, info.text_len, info.n_txps, info.n_keys
        self.perfect_hash = bool(info.perfect_hash)
        self.big_sa = bool(info.big_sa)     # int64 on disk (RapMapSAIndex<int64_t, ...>), unsigned 32-bit offsets on the device
        self._names = None
        self._lens = None

    @property
    def txp_names(self):
        if self._names is None:
            self._names = [lib().qm_index_txp_name(self._h, i).decode() for i in range(self.n_txps)]
        return self._names

    @property
    def txp_lens(self):
        if self._lens is None:
            self._lens = np.array([lib().qm_index_txp_len(self._h, i) for i in range(self.n_txps)], dtype=np.int64)
        return self._lens

    def arrays(self):
        """(text uint8[n], txpOffsets int64[T]) -- copies of rmi.seq / rmi.txpOffsets"""
        tp, tl, op, nt = C.c_void_p(), C.c_int64(), C.c_void_p(), C.c_int64()
        _check(lib().qm_index_arrays(self._h, C.byref(tp), C.byref(tl), C.byref(op), C.byref(nt)))
        text = np.ctypeslib.as_array(C.cast(tp, C.POINTER(C.c_uint8)), shape=(tl.value,)).copy()
        raw = np.ctypeslib.as_array(C.cast(op, C.POINTER(C.c_uint8)), shape=(nt.value * 4,)).copy()
        return text, raw.view("<u4").astype(np.int64)      # unsigned: a BigSA index has offsets beyond 2^31

    def raw(self, which):
        """zero-copy view of one of the library's own arrays of the open index (qm_index_raw): "sa" uint32[nSA], "hash" records
        {key u8, lb u4, ub u4} in file order (empty for a -p index), "complete_lens" uint32[T]; valid until close()"""
        code, dt = {"sa": (0, np.dtype("<u4")), "hash": (1, np.dtype([("key", "<u8"), ("lb", "<u4"), ("ub", "<u4")])), "complete_lens": (2, np.dtype("<u4"))}[which]
        p, n = C.c_void_p(), C.c_int64()
        _check(lib().qm_index_raw(self._h, code, C.byref(p), C.byref(n)))
        return _view(p, n.value, dt)

    def close(self):
        if self._h:
            lib().qm_index_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MapResult:
    __slots__ = ("hit_offsets", "hits", "counters", "n_hits", "map_kernel_ms", "total_ms")


class QuasiMapper:
    """One GPU context: the index replicated in HBM + work buffers.  `map_pairs` performs, for every
    pair, SACollector() x2 -> hitsToMappingsSimple() x2 -> mergeLeftRightHits() on one wavefront."""

    def __init__(self, index: QuasiIndex, device=0, debug=False, reuse_results=False, ph_compact=False, pair_kernel=True, wide_reads=False):
        """reuse_results: hit arrays of successive calls share one buffer (views valid until the next call) instead of
        a fresh allocation per batch, whose pages would have to be faulted in again every time -- for streaming callers
        that are done with a batch before they map the next (the CLI)"""
        self._reuse = bool(reuse_results)
        self._buf_hits = None
        self._buf_offs = None
        self.index = index
        self._h = C.c_void_p()
        # ph_compact: a -p index keeps the BooPHF / FrugalBooMap structure on the device instead of the expanded bucket table
        # pair_kernel=False: QM_CTX_NO_PAIR_KERNEL (paired reads of up to 128 characters take qm_lean_kernel alone); wide_reads: QM_CTX_WIDE_READS
        _check(lib().qm_ctx_create_ex(index._h, device, (1 if ph_compact else 0) | (0 if pair_kernel else 2) | (4 if wide_reads else 0), C.byref(self._h)))
        if debug:
            _check(lib().qm_ctx_set_debug(self._h, 1))
        self.device = device

    @property
    def device_bytes(self):
        return lib().qm_ctx_device_bytes(self._h)

    def _finish(self, n, n_hits, ctr, fetch=True):
        r = MapResult()
        r.n_hits = n_hits.value
        r.counters = ctr.as_dict()
        a, b = C.c_double(), C.c_double()
        _check(lib().qm_last_kernel_ms(self._h, C.byref(a), C.byref(b)))
        r.map_kernel_ms, r.total_ms = a.value, b.value
        if fetch and self._reuse:
            nh = max(r.n_hits, 0)
            if self._buf_hits is None or self._buf_hits.size < nh:
                self._buf_hits = np.empty(nh + nh // 4 + 1024, dtype=HIT_DTYPE)
            if self._buf_offs is None or self._buf_offs.size < n + 1:
                self._buf_offs = np.empty(n + 1 + n // 4, dtype=np.int64)
            r.hit_offsets = self._buf_offs[: n + 1]
            r.hits = self._buf_hits[:nh]
            _check(lib().qm_fetch_hits(self._h, r.hit_offsets.ctypes.data, r.hits.ctypes.data if r.n_hits else None))
        elif fetch:
            r.hit_offsets = np.zeros(n + 1, dtype=np.int64)
            r.hits = np.zeros(max(r.n_hits, 0), dtype=HIT_DTYPE)
            _check(lib().qm_fetch_hits(self._h, r.hit_offsets.ctypes.data, r.hits.ctypes.data if r.n_hits else None))
        else:
            r.hit_offsets = r.hits = None
        return r

    def map_pairs(self, seq1, off1, seq2, off2, opts=None):
        opts = opts or default_opts()
        seq1 = np.ascontiguousarray(seq1, dtype=np.uint8); off1 = np.ascontiguousarray(off1, dtype=np.int64)
        seq2 = np.ascontiguousarray(seq2, dtype=np.uint8); off2 = np.ascontiguousarray(off2, dtype=np.int64)
        n = len(off1) - 1
        if len(off2) - 1 != n:
            raise ValueError("left/right read counts differ")
        nh, ctr = C.c_int64(0), QmCounters()
        # numpy gives a non-null pointer even for empty arrays
        _check(lib().qm_map_pairs(self._h, C.byref(opts), n, seq1.ctypes.data or 1, off1.ctypes.data,
                                  seq2.ctypes.data or 1, off2.ctypes.data, C.byref(nh), C.byref(ctr)))
        return self._finish(n, nh, ctr)

    def map_reads(self, seq, off, opts=None):
        opts = opts or default_opts()
        seq = np.ascontiguousarray(seq, dtype=np.uint8); off = np.ascontiguousarray(off, dtype=np.int64)
        n = len(off) - 1
        nh, ctr = C.c_int64(0), QmCounters()
        _check(lib().qm_map_reads(self._h, C.byref(opts), n, seq.ctypes.data or 1, off.ctypes.data, C.byref(nh), C.byref(ctr)))
        return self._finish(n, nh, ctr)

    def map_device(self, n, d_seq1, d_off1, d_seq2, d_off2, max_read_len, opts=None, fetch=False):
        """Inputs are raw device pointers (ints) of arrays already resident in this GPU's HBM."""
        opts = opts or default_opts()
        nh, ctr = C.c_int64(0), QmCounters()
        _check(lib().qm_map_device(self._h, C.byref(opts), n, d_seq1, d_off1, d_seq2, d_off2, max_read_len,
                                   C.byref(nh), C.byref(ctr)))
        return self._finish(n, nh, ctr, fetch=fetch)

    def skipped(self):
        """qm_fetch_skipped: the reads of the last call that were skipped, not mapped -> (total, reads int64[], codes int32[]);
        code 1: longer than QM_MAX_LONG_READ_LEN characters, 2: interval lists beyond the scratch (max_interval above its default)"""
        tot = C.c_int64()
        _check(lib().qm_fetch_skipped(self._h, None, None, 0, C.byref(tot)))
        n = min(tot.value, 4096)
        reads = np.zeros(n, dtype=np.int64); codes = np.zeros(n, dtype=np.int32)
        if n:
            _check(lib().qm_fetch_skipped(self._h, reads.ctypes.data, codes.ctypes.data, n, C.byref(tot)))
        return tot.value, reads, codes

    def stat(self, which):
        """qm_ctx_stat: 0 stage-A relaunches of the last call, 1 list buffer capacity (words), 2 reads on the -s slow path,
        3 reads the lean stage-A kernel was launched over (-1: the general kernel ran), 4 reads it left to the general kernel,
        5 reads that were skipped (qm_fetch_skipped), 6 / 7 / 8 (-s) alignment questions beyond PERFECT chains / ksw2 alignments run for them / alignments the strip DP answered,
        9 pairs the pair kernel was launched over (-1: not used), 10 pairs it merged itself, 11 .. 14 why the reads of 4 were left: a character that is not
        A C G T / a window of k equal bases / an interval or a list beyond the kernel's lanes / hits on the other strand as well,
        15 reads with N's that the N-aware second pass of stage A mapped (not part of 4 or 11)"""
        v = C.c_int64()
        _check(lib().qm_ctx_stat(self._h, int(which), C.byref(v)))
        return v.value

    # ---- the reference's three entry points as calls of their own (include/qmap_mi355.h) ----
    def collect_reads(self, seq, off, opts=None):
        """SACollector::operator() for every read: (found uint8[n], int_offsets int64[n+1], intervals)"""
        opts = opts or default_opts()
        seq = np.ascontiguousarray(seq, dtype=np.uint8); off = np.ascontiguousarray(off, dtype=np.int64)
        n = len(off) - 1
        ni = C.c_int64(0)
        _check(lib().qm_collect_reads(self._h, C.byref(opts), n, seq.ctypes.data or 1, off.ctypes.data, C.byref(ni)))
        found = np.zeros(n + 1, dtype=np.uint8)
        _check(lib().qm_fetch_found(self._h, found.ctypes.data))
        offs, ints = self.intervals(n)
        return found[:n], offs, ints

    def hits_to_mappings(self, read_len, int_offsets, ints, opts=None):
        """hitsToMappingsSimple for every read from its SA-interval hits: (list_offsets int64[n+1], words uint64[])"""
        opts = opts or default_opts()
        read_len = np.ascontiguousarray(read_len, dtype=np.int32); int_offsets = np.ascontiguousarray(int_offsets, dtype=np.int64)
        ints = np.ascontiguousarray(ints, dtype=INTERVAL_DTYPE)
        n = len(read_len)
        nw = C.c_int64(0)
        _check(lib().qm_hits_to_mappings(self._h, C.byref(opts), n, read_len.ctypes.data or 1, int_offsets.ctypes.data,
                                         ints.ctypes.data if ints.size else None, C.byref(nw)))
        return self.read_lists(n)

    def read_lists(self, nreads):
        lo = np.zeros(nreads + 1, dtype=np.int64)
        _check(lib().qm_fetch_read_lists(self._h, lo.ctypes.data, None, 0))
        w = np.zeros(int(lo[-1]) + 1, dtype=np.uint64)
        _check(lib().qm_fetch_read_lists(self._h, lo.ctypes.data, w.ctypes.data, int(lo[-1])))
        return lo, w[: int(lo[-1])]

    def merge_lists(self, lo_l, w_l, lo_r, w_r, found_l, found_r, len_l, len_r, opts=None):
        """mergeLeftRightHits[Fuzzy] for every pair: MapResult (hits, counters) + .too_many uint8[n]"""
        opts = opts or default_opts()
        a = [np.ascontiguousarray(lo_l, dtype=np.int64), np.ascontiguousarray(w_l, dtype=np.uint64), np.ascontiguousarray(lo_r, dtype=np.int64),
             np.ascontiguousarray(w_r, dtype=np.uint64), np.ascontiguousarray(found_l, dtype=np.uint8), np.ascontiguousarray(found_r, dtype=np.uint8),
             np.ascontiguousarray(len_l, dtype=np.int32), np.ascontiguousarray(len_r, dtype=np.int32)]
        n = len(a[0]) - 1
        nh, ctr = C.c_int64(0), QmCounters()
        _check(lib().qm_merge_lists(self._h, C.byref(opts), n, *[x.ctypes.data if x.size else None for x in a], C.byref(nh), C.byref(ctr)))
        r = self._finish(n, nh, ctr)
        tm = np.zeros(n + 1, dtype=np.uint8)
        _check(lib().qm_fetch_too_many(self._h, tm.ctypes.data))
        r_too = tm[:n]
        return r, r_too

    def map_pairs_packed(self, seq1, off1, seq2, off2, opts=None):
        """map_pairs with the reads sent 2-bit packed (qm_pack_reads on the host, qm_map_pairs_packed): same result"""
        opts = opts or default_opts()
        a = [pack_2bit(seq1, off1), pack_2bit(seq2, off2)]
        n = len(a[0][1]) - 1
        nh, ctr = C.c_int64(0), QmCounters()
        _check(lib().qm_map_pairs_packed(self._h, C.byref(opts), n, a[0][0].ctypes.data, a[0][1].ctypes.data, a[0][2].ctypes.data, len(a[0][2]),
                                         a[1][0].ctypes.data, a[1][1].ctypes.data, a[1][2].ctypes.data, len(a[1][2]), C.byref(nh), C.byref(ctr)))
        return self._finish(n, nh, ctr)

    def map_pairs_prepacked(self, pk1, off1, exc1, pk2, off2, exc2, opts=None, fetch=True):
        """qm_map_pairs_packed on reads that are packed already (pack_2bit / pinned_copy): what a caller that keeps its batches 2-bit packed pays per call"""
        opts = opts or default_opts()
        n = len(off1) - 1
        nh, ctr = C.c_int64(0), QmCounters()
        _check(lib().qm_map_pairs_packed(self._h, C.byref(opts), n, pk1.ctypes.data, off1.ctypes.data, exc1.ctypes.data if len(exc1) else None, len(exc1),
                                         pk2.ctypes.data, off2.ctypes.data, exc2.ctypes.data if len(exc2) else None, len(exc2), C.byref(nh), C.byref(ctr)))
        return self._finish(n, nh, ctr, fetch=fetch)

    def map_reads_packed(self, seq, off, opts=None):
        opts = opts or default_opts()
        pk, off, exc = pack_2bit(seq, off)
        n = len(off) - 1
        nh, ctr = C.c_int64(0), QmCounters()
        _check(lib().qm_map_reads_packed(self._h, C.byref(opts), n, pk.ctypes.data, off.ctypes.data, exc.ctypes.data, len(exc), C.byref(nh), C.byref(ctr)))
        return self._finish(n, nh, ctr)

    def map_pairs_stages(self, seq1, off1, seq2, off2, opts=None, no_intervals=False, packed=False):
        """the three stages fused, every stage's output kept: MapResult of the merge (no caller-level bookkeeping).
        no_intervals: QM_STAGES_NO_INTERVALS (the stage view's interval arrays stay empty); packed: the reads go up 2-bit packed"""
        opts = opts or default_opts()
        seq1 = np.ascontiguousarray(seq1, dtype=np.uint8); off1 = np.ascontiguousarray(off1, dtype=np.int64)
        seq2 = np.ascontiguousarray(seq2, dtype=np.uint8); off2 = np.ascontiguousarray(off2, dtype=np.int64)
        n = len(off1) - 1
        nh, ctr = C.c_int64(0), QmCounters()
        fl = 1 if no_intervals else 0
        if packed:
            a = [pack_2bit(seq1, off1), pack_2bit(seq2, off2)]
            _check(lib().qm_map_pairs_stages_packed(self._h, C.byref(opts), C.c_int64(n), C.c_void_p(a[0][0].ctypes.data), C.c_void_p(a[0][1].ctypes.data),
                                                    C.c_void_p(a[0][2].ctypes.data if len(a[0][2]) else None), C.c_int64(len(a[0][2])),
                                                    C.c_void_p(a[1][0].ctypes.data), C.c_void_p(a[1][1].ctypes.data),
                                                    C.c_void_p(a[1][2].ctypes.data if len(a[1][2]) else None), C.c_int64(len(a[1][2])), C.c_uint32(fl), C.byref(nh), C.byref(ctr)))
        elif fl:
            _check(lib().qm_map_pairs_stages_ex(self._h, C.byref(opts), C.c_int64(n), C.c_void_p(seq1.ctypes.data or 1), C.c_void_p(off1.ctypes.data),
                                                C.c_void_p(seq2.ctypes.data or 1), C.c_void_p(off2.ctypes.data), C.c_uint32(fl), C.byref(nh), C.byref(ctr)))
        else:
            _check(lib().qm_map_pairs_stages(self._h, C.byref(opts), n, seq1.ctypes.data or 1, off1.ctypes.data,
                                             seq2.ctypes.data or 1, off2.ctypes.data, C.byref(nh), C.byref(ctr)))
        return self._finish(n, nh, ctr)

    def fetch_stages(self, pinned=True):
        """qm_fetch_stages after map_pairs_stages: every stage's output in one download, compacted on the device.  Returns a dict of
        numpy views into an arena this mapper keeps (valid until the next fetch_stages): iv_off / iv / found per READ (2u left,
        2u + 1 right), list_off / words per read, hit_off / hits / too_many per pair."""
        need = C.c_int64(0)
        _check(lib().qm_stage_bytes(self._h, C.byref(need)))
        if getattr(self, "_arena_cap", 0) < need.value:
            if getattr(self, "_arena", None) and self._arena_pinned:
                lib().qm_pinned_free(self._arena)
            self._arena_cap = need.value + need.value // 2 + 4096
            self._arena_pinned = bool(pinned)
            if pinned:
                self._arena = lib().qm_pinned_alloc(self._arena_cap)
                if not self._arena:
                    raise QmError("out of page-locked host memory")
            else:
                self._arena_np = np.zeros(self._arena_cap, dtype=np.uint8)
                self._arena = self._arena_np.ctypes.data
        v = QmStageView()
        _check(lib().qm_fetch_stages(self._h, C.c_void_p(self._arena), self._arena_cap, C.byref(v)))
        nr, n = v.n_reads, v.n_units
        out = {"n_units": n, "n_reads": nr}
        out["iv_off"] = _view(v.iv_off, nr + 1, np.int64); out["iv"] = _view(v.iv, int(out["iv_off"][-1]), INTERVAL_DTYPE)
        out["found"] = _view(v.found, nr, np.uint8)
        out["list_off"] = _view(v.list_off, nr + 1, np.int64); out["words"] = _view(v.words, int(out["list_off"][-1]), np.uint64)
        out["hit_off"] = _view(v.hit_off, n + 1, np.int64); out["hits"] = _view(v.hits, int(out["hit_off"][-1]), HIT_DTYPE)
        out["too_many"] = _view(v.too_many, n, np.uint8)
        return out

    def intervals(self, n):
        """fwdSAInts / rcSAInts kept by the last call (fused calls: needs debug=True)."""
        offs = np.zeros(n + 1, dtype=np.int64)
        _check(lib().qm_fetch_intervals(self._h, offs.ctypes.data, None, 0))
        ints = np.zeros(int(offs[-1]), dtype=INTERVAL_DTYPE)
        if ints.size:
            _check(lib().qm_fetch_intervals(self._h, offs.ctypes.data, ints.ctypes.data, ints.size))
        return offs, ints

    def close(self):
        if self._h:
            if getattr(self, "_arena", None) and getattr(self, "_arena_pinned", False):
                lib().qm_pinned_free(self._arena)
            self._arena = None; self._arena_cap = 0
            lib().qm_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


PACK_EXC_DTYPE = np.dtype([("pos", "<u4"), ("ch", "<u4")])


def pinned_copy(a):
    """a copy of the array in page-locked host memory (qm_pinned_alloc; kept for the life of the process): DMA source of the host-buffer calls"""
    a = np.ascontiguousarray(a)
    p = lib().qm_pinned_alloc(max(64, a.nbytes))
    if not p:
        raise QmError("qm_pinned_alloc(%d) failed" % a.nbytes)
    v = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(64, a.nbytes),))[: a.nbytes].view(a.dtype).reshape(a.shape)
    v[...] = a
    return v


def pack_2bit(seq, off):
    """qm_pack_reads: (packed uint8[], off int64[n+1], exceptions {pos, ch}[]) of a batch of reads -- four characters to a byte,
    read i from byte (off[i] >> 2) + i on, everything that is not upper-case A C G T as an exception"""
    seq = np.ascontiguousarray(seq, dtype=np.uint8); off = np.ascontiguousarray(off, dtype=np.int64)
    n = len(off) - 1
    pk = np.zeros((int(off[-1]) >> 2) + n + 8, dtype=np.uint8)
    cap = int(off[-1]) + 16
    exc = np.zeros(cap, dtype=PACK_EXC_DTYPE)
    ne = C.c_int64(0)
    rc = lib().qm_pack_reads(seq.ctypes.data or 1, off.ctypes.data, n, pk.ctypes.data, exc.ctypes.data, cap, C.byref(ne))
    if rc != 0:
        raise QmError(lib().qm_io_last_error().decode())
    return pk, off, exc[: ne.value].copy()


def unpack_2bit(pk, off, exc):
    """the inverse, in numpy (tests): the characters the device's unpack kernels write"""
    off = np.asarray(off, dtype=np.int64); n = len(off) - 1
    out = np.zeros(int(off[-1]), dtype=np.uint8)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    for i in range(n):
        o, e = int(off[i]), int(off[i + 1])
        b = pk[(o >> 2) + i:(o >> 2) + i + (e - o + 3) // 4]
        codes = ((b[:, None] >> (2 * np.arange(4))[None, :]) & 3).reshape(-1)[: e - o]
        out[o:e] = lut[codes]
    out[exc["pos"]] = exc["ch"].astype(np.uint8)
    return out


class _Mem:
    """array-interface carrier: np.asarray(_Mem(...)) is a zero-copy view of foreign memory in a few microseconds
    (np.ctypeslib.as_array builds a new ctypes array type per shape: ~1 ms per view, which at 2^18-pair batches was a third
    of the stream's wall time)"""
    __slots__ = ("__array_interface__",)

    def __init__(self, addr, count, dt):
        self.__array_interface__ = {"data": (addr, False), "shape": (count,), "typestr": dt.str, "version": 3}
        if dt.fields:
            self.__array_interface__["descr"] = dt.descr


def _view(p, count, dt):
    """zero-copy numpy view of `count` elements of dtype dt at the address held by a ctypes pointer / integer"""
    addr = p.value if hasattr(p, "value") else p
    if count == 0 or not addr:
        return np.zeros(0, dtype=dt)
    return np.asarray(_Mem(int(addr), int(count), np.dtype(dt)))


class ReadBatch:
    """one chunk handed out by FastxReader: packed sequences/names (numpy views, valid until the next chunk)"""
    pass


class FastxReader:
    """FASTA/FASTQ(.gz) ingest through the library's native reader (qm_reader_*): the packed batches that
    QuasiMapper.map_pairs / map_reads take.  path2=None for single-end input."""

    def __init__(self, path1, path2=None, threads=None):
        self._h = C.c_void_p()
        rc = lib().qm_reader_open(path1.encode(), path2.encode() if path2 else None,
                                  int(threads or min(16, os.cpu_count() or 1)), C.byref(self._h))
        if rc != 0:
            raise QmError(lib().qm_io_last_error().decode())
        self.paired = path2 is not None

    def chunks(self, max_units):
        L = lib()
        n = C.c_int64()
        ptr = [C.c_void_p() for _ in range(8)]
        while True:
            rc = L.qm_reader_next(self._h, int(max_units), C.byref(n), *[C.byref(x) for x in ptr])
            if rc != 0:
                raise QmError(L.qm_io_last_error().decode())
            if n.value == 0:
                return
            b = ReadBatch(); b.n = n.value

            arr = _view
            b.off1 = arr(ptr[1], b.n + 1, np.int64); b.seq1 = arr(ptr[0], int(b.off1[-1]), np.uint8)
            b.name_off1 = arr(ptr[3], b.n + 1, np.int64); b.names1 = arr(ptr[2], int(b.name_off1[-1]), np.uint8)
            if self.paired:
                b.off2 = arr(ptr[5], b.n + 1, np.int64); b.seq2 = arr(ptr[4], int(b.off2[-1]), np.uint8)
                b.name_off2 = arr(ptr[7], b.n + 1, np.int64); b.names2 = arr(ptr[6], int(b.name_off2[-1]), np.uint8)
            yield b

    def close(self):
        if self._h:
            lib().qm_reader_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def reserve_stream_memory(nbytes=512 << 20):
    """qm_stream_reserve: pin the process-wide pool the streams' slots come out of, in the background (returns at once).  Call it
    early -- before the index is uploaded -- so that pinning (5.5 GB/s) runs under other work."""
    _check(lib().qm_stream_reserve(int(nbytes)))


class MappedStream:
    """FASTA/FASTQ files -> mapped batches through the library's pipelined stream (qm_stream_*): ingest workers packing batches
    into pinned slots, two device contexts per device, results in pinned memory.  `device` may be one id or a list of ids
    (consecutive batches go to different devices; batches still come back in input order).  Iterating yields ReadBatch
    objects that also carry hit_offsets / hits / counters / device; every array is a zero-copy view that stays valid until
    the next batch is taken.  names=False: read names are not kept (hits-only callers)."""

    def __init__(self, index: QuasiIndex, path1, path2=None, opts=None, device=0, batch_units=1 << 20, threads=None, ph_compact=False,
                 names=True):
        self._h = C.c_void_p()
        self.paired = path2 is not None
        self.names = bool(names)
        opts = opts or default_opts()
        devs = [int(d) for d in device] if isinstance(device, (list, tuple)) else [int(device)]
        self.devices = devs
        darr = (C.c_int32 * len(devs))(*devs)
        rc = lib().qm_stream_open_ex(index._h, darr, len(devs), 1 if ph_compact else 0, C.byref(opts), os.fsencode(path1),
                                     os.fsencode(path2) if path2 else None, int(batch_units),
                                     int(threads or min(32, os.cpu_count() or 1)), 0 if names else 1, C.byref(self._h))
        if rc != 0:
            raise QmError("qm_stream_open failed (%d): %s" % (rc, lib().qm_stream_last_error().decode(errors="replace")))
        self._index = index

    def __iter__(self):
        L = lib()
        sb = QmStreamBatch()

        arr = _view
        while True:
            rc = L.qm_stream_next(self._h, C.byref(sb))
            if rc != 0:
                raise QmError("qm_stream_next failed (%d): %s" % (rc, L.qm_stream_last_error().decode(errors="replace")))
            if sb.n_units == 0:
                return
            b = ReadBatch(); b.n = n = sb.n_units
            b.off1 = arr(sb.off1, n + 1, np.int64); b.seq1 = arr(sb.seq1, int(b.off1[-1]), np.uint8)
            if self.names:
                b.name_off1 = arr(sb.name_off1, n + 1, np.int64); b.names1 = arr(sb.names1, int(b.name_off1[-1]), np.uint8)
            if self.paired:
                b.off2 = arr(sb.off2, n + 1, np.int64); b.seq2 = arr(sb.seq2, int(b.off2[-1]), np.uint8)
                if self.names:
                    b.name_off2 = arr(sb.name_off2, n + 1, np.int64); b.names2 = arr(sb.names2, int(b.name_off2[-1]), np.uint8)
            b.device = sb.device
            b.hit_offsets = arr(sb.hit_offsets, n + 1, np.int64)
            b.n_hits = sb.n_hits
            b.hits = arr(sb.hits, sb.n_hits, HIT_DTYPE) if sb.n_hits else np.zeros(0, dtype=HIT_DTYPE)
            b.counters = sb.counters.as_dict()
            b.gpu_ms = sb.gpu_ms
            yield b

    def stats(self):
        """seconds: read_s = open to the last batch packed (wall), map_s / fetch_s = upload + kernels / download summed over the
        contexts, parse_cpu_s / copy_cpu_s = the ingest workers' task time summed over the workers"""
        a = (C.c_double * 13)()
        _check(lib().qm_stream_stats_ex(self._h, a, 13))
        return dict(zip(("read_s", "map_s", "fetch_s", "caller_wait_s", "open_s", "alloc_s", "first_batch_s", "parse_cpu_s", "copy_cpu_s",
                         "inflate_s", "bytes_parsed", "last_mapped_s", "packed_batches"), [float(x) for x in a]))

    def close(self):
        if self._h:
            lib().qm_stream_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _take_buf(p, n):
    try:
        return C.string_at(p, n.value)
    finally:
        lib().qm_buf_free(p)


def sam_header_text(index: "QuasiIndex") -> bytes:
    p = C.c_void_p(); n = C.c_int64()
    _check(lib().qm_sam_header(index._h, C.byref(p), C.byref(n)))
    return _take_buf(p, n)


def _sam_batch_args(batch, hit_offsets, hits):
    ho = np.ascontiguousarray(hit_offsets, dtype=np.int64)
    hh = np.ascontiguousarray(hits)
    paired = getattr(batch, "seq2", None) is not None
    keep = [np.ascontiguousarray(x) for x in (batch.names1, batch.name_off1, batch.seq1, batch.off1)]
    if paired:
        keep += [np.ascontiguousarray(x) for x in (batch.names2, batch.name_off2, batch.seq2, batch.off2)]
    args = [C.c_void_p(a.ctypes.data) for a in keep] + ([C.c_void_p(0)] * 4 if not paired else [])
    if not keep[2].size:                       # a zero-length sequence array still needs a non-null pointer
        args[2] = C.c_void_p(ho.ctypes.data)
    args += [C.c_void_p(ho.ctypes.data), C.c_void_p(hh.ctypes.data if hh.size else ho.ctypes.data)]
    return args, (keep, ho, hh)


class SamWriter:
    """qm_sam_writer_*: SAM records of a run onto one file descriptor; put() formats a batch and returns while the writer's
    own thread writes the previous one."""

    def __init__(self, index: "QuasiIndex", fd, max_num_hits=200, threads=None, gzip=False, level=1):
        """gzip=True: `-x` -- every part of a batch becomes a gzip member compressed by its own worker (a valid .gz stream)"""
        self._h = C.c_void_p()
        self._index = index
        flags = (1 | (int(level) & 15) << 8) if gzip else 0
        rc = lib().qm_sam_writer_open_ex(index._h, int(fd), int(max_num_hits), int(threads or min(16, os.cpu_count() or 1)), flags, C.byref(self._h))
        if rc != 0:
            raise QmError(lib().qm_io_last_error().decode())

    def header(self):
        """the SAM header, through the writer's queue (compressed like the records when gzip=True)"""
        rc = lib().qm_sam_writer_header(self._h)
        if rc != 0:
            raise QmError(lib().qm_io_last_error().decode())

    def put(self, batch, hit_offsets, hits):
        args, keep = _sam_batch_args(batch, hit_offsets, hits)
        rc = lib().qm_sam_writer_put(self._h, int(batch.n), *args)
        if rc != 0:
            raise QmError(lib().qm_io_last_error().decode())

    def close(self):
        """drains the writer; returns the bytes written"""
        if not self._h:
            return 0
        nb = C.c_int64()
        rc = lib().qm_sam_writer_close(self._h, C.byref(nb))
        self._h = C.c_void_p()
        if rc != 0:
            raise QmError(lib().qm_io_last_error().decode())
        return nb.value


def sam_records_text(index: "QuasiIndex", batch, hit_offsets, hits, max_num_hits=200, threads=None, fd=None):
    """SAM records of one mapped ReadBatch (paired when the batch has mates), formatted by the library.
    fd=None: returns the text; otherwise writes it to that file descriptor and returns the byte count."""
    def vp(a):
        return C.c_void_p(np.ascontiguousarray(a).ctypes.data) if a is not None and len(a) else C.c_void_p(0)
    ho = np.ascontiguousarray(hit_offsets, dtype=np.int64)
    hh = np.ascontiguousarray(hits)
    paired = getattr(batch, "seq2", None) is not None
    keep = [np.ascontiguousarray(x) for x in (batch.names1, batch.name_off1, batch.seq1, batch.off1)]
    if paired:
        keep += [np.ascontiguousarray(x) for x in (batch.names2, batch.name_off2, batch.seq2, batch.off2)]
    args = [C.c_void_p(a.ctypes.data) for a in keep] + ([C.c_void_p(0)] * 4 if not paired else [])
    # a zero-length sequence array still needs a non-null pointer
    if not keep[2].size:
        args[2] = C.c_void_p(ho.ctypes.data)
    nthr = int(threads or min(16, os.cpu_count() or 1))
    if fd is not None:
        nb = C.c_int64()
        rc = lib().qm_sam_write(index._h, int(batch.n), *args, C.c_void_p(ho.ctypes.data),
                                C.c_void_p(hh.ctypes.data if hh.size else ho.ctypes.data), int(max_num_hits), nthr, int(fd),
                                C.byref(nb))
        if rc != 0:
            raise QmError(lib().qm_io_last_error().decode())
        return nb.value
    p = C.c_void_p(); n = C.c_int64()
    rc = lib().qm_sam_records(index._h, int(batch.n), *args, C.c_void_p(ho.ctypes.data),
                              C.c_void_p(hh.ctypes.data if hh.size else ho.ctypes.data), int(max_num_hits),
                              int(threads or min(16, os.cpu_count() or 1)), C.byref(p), C.byref(n))
    if rc != 0:
        raise QmError(lib().qm_io_last_error().decode())
    return _take_buf(p, n)
