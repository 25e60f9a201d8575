"""tests/emu/emu.py -- ctypes face of the TEST-ONLY lane emulation (libqm_emu.so)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libqm_emu.so")
_SRC = [os.path.join(_HERE, "qm_emu.cpp"),
        os.path.join(_HERE, "../../rapmap_amd/csrc/qm_mapper.inl"),
        os.path.join(_HERE, "../../rapmap_amd/csrc/qm_lean.inl"),
        os.path.join(_HERE, "../../rapmap_amd/csrc/qm_duo.inl"),
        os.path.join(_HERE, "../../rapmap_amd/csrc/qm_wave.h"),
        os.path.join(_HERE, "../../rapmap_amd/csrc/qm_phflat.h"),
        os.path.join(_HERE, "../../rapmap_amd/csrc/qm_sel.inl"),
        os.path.join(_HERE, "../../rapmap_amd/csrc/qm_selpack.inl")]

HIT_DTYPE = np.dtype([
    ("tid", "<u4"), ("pos", "<i4"), ("mate_pos", "<i4"), ("frag_len", "<u4"),
    ("read_len", "<u4"), ("mate_len", "<u4"),
    ("fwd", "u1"), ("mate_is_fwd", "u1"), ("is_paired", "u1"), ("mate_status", "u1"),
    ("aln_score", "<i4"),
])
INT_DTYPE = np.dtype([("begin", "<i4"), ("end", "<i4"), ("len", "<u4"), ("query_pos", "<u4"),
                      ("query_rc", "u1"), ("list", "u1"), ("pad", "<u2")])


class QmOpts(C.Structure):
    _fields_ = [("sensitive", C.c_int32), ("strict_check", C.c_int32), ("max_num_hits", C.c_int32),
                ("no_orphans", C.c_int32), ("no_dovetail", C.c_int32), ("fuzzy", C.c_int32),
                ("max_interval", C.c_int32), ("sel_aln", C.c_int32), ("quasi_cov", C.c_double),
                ("hard_filter", C.c_int32), ("match_score", C.c_int32), ("mismatch_penalty", C.c_int32), ("gap_open", C.c_int32),
                ("gap_extend", C.c_int32), ("dp_bandwidth", C.c_int32), ("max_mmp_extension", C.c_int32), ("aln_policy", C.c_int32),
                ("min_score_fraction", C.c_double), ("consensus_slack", C.c_double)]


def default_opts(**kw):
    o = QmOpts(1, 1, 200, 0, 0, 0, 1000, 0, 0.0, 0, 2, -4, 4, 2, 15, 7, 0, 0.65, 0.2)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def build():
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unused", "-o", _LIB, _SRC[0]])


def _lib():
    if not os.path.exists(_LIB) or any(os.path.getmtime(_LIB) < os.path.getmtime(s) for s in _SRC):
        build()
    lib = C.CDLL(_LIB)
    lib.qe_slots_cap.restype = C.c_uint64
    lib.qe_slots_cap.argtypes = [C.c_int64]
    return lib


class Emu:
    def __init__(self, ix, buckets=None):
        """buckets: override the number of hash buckets (power of two) to force bucket overflow chains"""
        self.lib = _lib()
        self.ix = ix
        n = ix.text.size
        self.text = np.zeros(n + 128, dtype=np.uint8)
        self.text[:n] = ix.text
        self.n = n
        self.SA = np.ascontiguousarray(ix.SA, dtype=np.uint32)     # unsigned 32-bit on the device, also for a BigSA index (int64 on disk)
        self.sainfo = np.zeros(self.SA.size * 2, dtype=np.uint32)
        self.cap = int(buckets) if buckets else int(self.lib.qe_slots_cap(ix.hkeys.size))
        self.slots = np.zeros(self.cap * 8 + 8, dtype=np.uint64)   # cap buckets of 64 bytes
        off = np.ascontiguousarray(ix.txpOffsets, dtype=np.uint32)
        self.txp_off = off
        self.txp_len = np.ascontiguousarray(ix.txpLens, dtype=np.int32)
        hk = np.ascontiguousarray(ix.hkeys, dtype=np.uint64)
        hl = np.ascontiguousarray(ix.hlb, dtype=np.uint32)
        hu = np.ascontiguousarray(ix.hub, dtype=np.uint32)
        self.ph = None
        if getattr(ix, "perfect", False):
            # perfect-hash index: the emulated device code walks the BooPHF levels itself
            from oracle import q5ph
            boo = q5ph.BooPHF(os.path.join(ix.dir, "hash_info.bph"))
            data, lens, ovf = q5ph.read_val(os.path.join(ix.dir, "hash_info.val"), big=bool(ix.big))
            tab, words, ranks = [], [], []
            wo = ro = 0
            for (size, w, r), dom in zip(boo.levels, boo.domains):
                tab += [dom, wo, ro]; words.append(w); ranks.append(r); wo += w.size; ro += r.size
            self._ph_keep = [np.ascontiguousarray(np.concatenate(words), dtype=np.uint64),
                             np.ascontiguousarray(np.concatenate(ranks) if ro else np.zeros(1), dtype=np.uint64),
                             np.array(tab, dtype=np.uint64), np.ascontiguousarray(data, dtype=np.uint32),
                             np.ascontiguousarray(lens, dtype=np.uint8),
                             np.array([x for kv in ovf.items() for x in kv] or [0, 0], dtype=np.uint32),
                             np.array([x for kv in boo.final.items() for x in kv] or [0, 0], dtype=np.uint64)]
            k = self._ph_keep
            self.lib.qe_ph_create.restype = C.c_void_p
            self.ph = self.lib.qe_ph_create(C.c_void_p(k[0].ctypes.data), C.c_void_p(k[1].ctypes.data), C.c_void_p(k[2].ctypes.data),
                                            C.c_int(boo.nb_levels), C.c_void_p(k[3].ctypes.data), C.c_void_p(k[4].ctypes.data),
                                            C.c_uint64(boo.nelem), C.c_uint64(boo.lastbitsetrank), C.c_void_p(k[5].ctypes.data),
                                            C.c_int64(len(ovf)), C.c_void_p(k[6].ctypes.data), C.c_int64(len(boo.final)),
                                            C.c_uint64(wo), C.c_uint64(ro), C.c_int(ix.k), C.c_void_p(self.text.ctypes.data),
                                            C.c_int64(self.n), C.c_void_p(self.SA.ctypes.data), C.c_int64(self.SA.size))
            assert self.ph, "rank samples of hash_info.bph do not match its bit arrays"
        self.lib.qe_flatten(C.c_void_p(self.SA.ctypes.data), C.c_int64(self.SA.size), C.c_void_p(off.ctypes.data),
                            C.c_int64(off.size), C.c_void_p(self.sainfo.ctypes.data), C.c_void_p(hk.ctypes.data),
                            C.c_void_p(hl.ctypes.data), C.c_void_p(hu.ctypes.data), C.c_int64(hk.size),
                            C.c_void_p(self.slots.ctypes.data), C.c_uint64(self.cap), C.c_int(ix.k))

    def map(self, seq1, off1, seq2=None, off2=None, opts=None, ns=2):
        opts = opts or default_opts()
        nunits = len(off1) - 1
        pad = np.zeros(8, dtype=np.uint8)                        # reads are fetched four characters at a time
        seq1 = np.concatenate([np.asarray(seq1, dtype=np.uint8), pad]); off1 = np.ascontiguousarray(off1, dtype=np.int64)
        paired = seq2 is not None
        if paired:
            seq2 = np.concatenate([np.asarray(seq2, dtype=np.uint8), pad]); off2 = np.ascontiguousarray(off2, dtype=np.int64)
        ho = np.zeros(nunits + 1, dtype=np.int64); io = np.zeros(nunits + 1, dtype=np.int64)
        ctr = np.zeros(6, dtype=np.uint64)
        hp = C.c_void_p(); ip = C.c_void_p(); st = C.c_int(0)
        rc = self.lib.qe_map(C.c_int(self.ix.k), C.c_void_p(self.text.ctypes.data), C.c_int64(self.n),
                             C.c_void_p(self.SA.ctypes.data), C.c_int64(self.SA.size),
                             C.c_void_p(self.sainfo.ctypes.data), C.c_void_p(self.slots.ctypes.data),
                             C.c_uint64(self.cap - 1), C.byref(opts), C.c_int64(nunits),
                             C.c_void_p(seq1.ctypes.data), C.c_void_p(off1.ctypes.data),
                             C.c_void_p(seq2.ctypes.data if paired else None),
                             C.c_void_p(off2.ctypes.data if paired else None), C.c_int(ns), C.c_void_p(self.ph),
                             C.c_void_p(ho.ctypes.data), C.byref(hp), C.c_void_p(ctr.ctypes.data),
                             C.c_void_p(io.ctypes.data), C.byref(ip), C.byref(st),
                             C.c_void_p(self.txp_off.ctypes.data), C.c_void_p(self.txp_len.ctypes.data))
        assert rc == 0
        tot = int(ho[-1])
        hits = np.ctypeslib.as_array(C.cast(hp, C.POINTER(C.c_uint8)), shape=((tot + 1) * 32,))[: tot * 32].copy().view(HIT_DTYPE)
        ti = int(io[-1])
        ints = np.ctypeslib.as_array(C.cast(ip, C.POINTER(C.c_uint8)), shape=((ti + 1) * 20,))[: ti * 20].copy().view(INT_DTYPE)
        self.lib.qe_free(hp); self.lib.qe_free(ip)

        class R:
            pass
        r = R()
        r.hit_offsets, r.hits, r.int_offsets, r.ints, r.status = ho, hits, io, ints, st.value
        r.counters = dict(zip(["peHits", "seHits", "totHits", "numReads", "tooManyHits", "mappedUnits"], [int(x) for x in ctr]))
        return r
