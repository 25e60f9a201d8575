"""Config-2-size index cross-check (CPU only).  The GENCODE-like synthetic transcriptome of the bench (40 000
genes -> ~152 k transcripts, 2.58e8 characters, 7.97e7 31-mers) is indexed by the product's own builder (qm_build_index), then
 (1) every array the product's loader (qm_index_open: mmap + views) holds is compared with what the oracle's independent numpy
     reader (oracle/q5.py) parses out of the same files: text, transcript offsets / lengths / names, complete lengths, the suffix
     array, the dense hash's records in file order;
 (2) hash.bin goes through the REFERENCE's container -- spp::sparse_hash_map::unserialize compiled in place (oracle/_ref, the
     harness of tests/test_oracle_ref.py) -- and its find() must return every one of the records with the same interval, and nothing
     for absent keys;
 (3) the suffix array is verified against the text itself on a large random sample (adjacent suffixes in order), and every hash
     record's interval against the suffix array (first / last suffix carry the k-mer, the neighbours outside do not).
usage: python profiles/r04/big_index_crosscheck.py [genes=40000] [workdir=/tmp/qmap_bigidx]   -> JSON line on stdout"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

import rapmap_amd as ra
from rapmap_amd import synth
from oracle import q5

genes = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
work = sys.argv[2] if len(sys.argv) > 2 else "/tmp/qmap_bigidx"
os.makedirs(work, exist_ok=True)
idx = os.path.join(work, "idx_g%d" % genes)
out = {"genes": genes}
t = time.time()
if not os.path.exists(os.path.join(idx, "hash.bin")):
    names, txps = synth.make_transcriptome(genes, seed=42)
    fa = os.path.join(work, "t.fa"); synth.write_fasta(fa, names, txps); del names, txps
    ra.build_index(fa, idx, k=31, threads=os.cpu_count() or 1); os.remove(fa)
out["build_s"] = round(time.time() - t, 1)

t = time.time()
ox = q5.load(idx)                       # the oracle's reader
qi = ra.QuasiIndex(idx)                 # the product's loader
text, offs = qi.arrays()
chk = {}
chk["text"] = bool(np.array_equal(text, ox.text)); chk["txp_offsets"] = bool(np.array_equal(offs, ox.txpOffsets.astype(np.int64)))
chk["txp_lens"] = bool(np.array_equal(qi.txp_lens, ox.txpLens)); chk["txp_names"] = qi.txp_names == list(ox.names)
chk["complete_lens"] = bool(np.array_equal(qi.raw("complete_lens"), ox.completeLens))
sa = qi.raw("sa"); chk["sa"] = bool(np.array_equal(sa, ox.SA.astype(np.uint32)))
h = qi.raw("hash")
chk["hash_keys"] = bool(np.array_equal(h["key"], ox.hkeys)); chk["hash_lb"] = bool(np.array_equal(h["lb"], ox.hlb.astype(np.uint32)))
chk["hash_ub"] = bool(np.array_equal(h["ub"], ox.hub.astype(np.uint32)))
out.update(n_txps=int(qi.n_txps), text_len=int(qi.text_len), n_keys=int(qi.n_keys), n_sa=int(sa.size), loader_vs_numpy_reader=chk,
           compare_s=round(time.time() - t, 1))

# (3) the arrays against each other
t = time.time()
rng = np.random.default_rng(1)
n = text.size
i = rng.integers(0, sa.size - 1, size=400_000)
W = 48
def pref(pos):
    m = np.minimum(pos[:, None].astype(np.int64) + np.arange(W)[None, :], n - 1)
    a = text[m].copy(); a[(pos[:, None].astype(np.int64) + np.arange(W)[None, :]) >= n] = 0
    return a
a = pref(sa[i]); b = pref(sa[i + 1])
neq = a != b
first = np.where(neq.any(1), neq.argmax(1), W)
rows = np.arange(i.size)
ordered = (first == W) | (a[rows, np.minimum(first, W - 1)] < b[rows, np.minimum(first, W - 1)])
out["sa_adjacent_suffixes_in_order"] = {"sampled": int(i.size), "ok": int(ordered.sum())}
j = rng.integers(0, h.size, size=200_000)
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
inside = ok0 & ok1 & (w0 == h["key"][j]) & (w1 == h["key"][j])
wb, okb = kmer_at(sa[np.maximum(lb - 1, 0)]); wa, oka = kmer_at(sa[np.minimum(ub, sa.size - 1)])
outside = ((lb == 0) | ~okb | (wb != h["key"][j])) & ((ub == sa.size) | ~oka | (wa != h["key"][j]))
out["hash_intervals_vs_suffix_array"] = {"sampled": int(j.size), "ends_carry_the_kmer": int(inside.sum()), "neighbours_do_not": int(outside.sum())}
out["self_check_s"] = round(time.time() - t, 1)

# (2) the reference's container
t = time.time()
refso = os.path.join(ROOT, "oracle", "_ref", "libqm_ref.so")
if os.path.exists(refso):
    R = C.CDLL(refso)
    R.ref_spp_load.restype = C.c_void_p; R.ref_spp_load.argtypes = [C.c_char_p]
    R.ref_spp_size.restype = C.c_int64; R.ref_spp_size.argtypes = [C.c_void_p]
    R.ref_spp_find.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    R.ref_spp_free.argtypes = [C.c_void_p]
    hd = R.ref_spp_load(os.path.join(idx, "hash.bin").encode())
    assert hd, "the reference's sparse_hash_map could not unserialize hash.bin"
    size = int(R.ref_spp_size(hd))
    keys = np.ascontiguousarray(h["key"]); K = keys.size
    found = np.zeros(K, np.uint8); fl = np.zeros(K, np.int32); fu = np.zeros(K, np.int32)
    R.ref_spp_find(hd, keys.ctypes.data, K, found.ctypes.data, fl.ctypes.data, fu.ctypes.data)
    ok = int(((found == 1) & (fl.view(np.uint32) == h["lb"]) & (fu.view(np.uint32) == h["ub"])).sum())
    absent = rng.integers(0, 1 << 62, size=2_000_000, dtype=np.uint64)
    absent = absent[~np.isin(absent, keys)]
    fa_ = np.zeros(absent.size, np.uint8); x = np.zeros(absent.size, np.int32); y = np.zeros(absent.size, np.int32)
    R.ref_spp_find(hd, absent.ctypes.data, absent.size, fa_.ctypes.data, x.ctypes.data, y.ctypes.data)
    R.ref_spp_free(hd)
    out["reference_container"] = {"unserialized_size": size, "records": int(K), "found_with_the_same_interval": ok,
                                  "absent_keys_probed": int(absent.size), "absent_keys_found": int(fa_.sum())}
else:
    out["reference_container"] = "oracle/_ref/libqm_ref.so not built"
out["reference_container_s"] = round(time.time() - t, 1)
qi.close()
print(json.dumps(out))
