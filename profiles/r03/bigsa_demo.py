"""BigSA on a text that really needs it: > 2^31 characters ("a synthetic > 2^31-char index opens and maps
(small read set) on the GPU").

A pan-genome-like text: HAPS near-copies (SUBS substitutions per character) of a random base sequence, cut into transcripts
of TXP characters -- 2.2 G characters keep the number of distinct 31-mers (and so the dense table) small enough for the box
while every k-mer's SA interval is ~HAPS wide and lies anywhere in [0, 2.2 G), i.e. beyond 2^31 for a few percent of them, and
the last haplotypes' text positions are beyond 2^31 too.

  python profiles/r03/bigsa_demo.py [--chars 2.2e9] [--pairs 200000] [--dir /tmp/bigsa] [--no-gpu]

Steps: write the FASTA, build the index with the product's indexer (int64 form chosen by the text length, no forcing), open it
(narrowed to unsigned 32-bit), map simulated pairs on the GPU, map the same pairs with the oracle built with IndexT = int64_t,
compare bit for bit (hits, counters, SA-interval records), report how many interval bounds / text positions lay beyond 2^31."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chars", type=float, default=2.2e9)
    ap.add_argument("--haps", type=int, default=64)
    ap.add_argument("--txp", type=int, default=10000)
    ap.add_argument("--subs", type=float, default=0.001)
    ap.add_argument("--pairs", type=int, default=200000)
    ap.add_argument("--dir", default="/tmp/bigsa")
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--no-gpu", action="store_true")
    ap.add_argument("--force-big", action="store_true", help="small --chars rehearsal: QM_FORCE_BIGSA")
    a = ap.parse_args()
    import rapmap_amd as ra
    from oracle import oracle, q5
    os.makedirs(a.dir, exist_ok=True)
    out = {"chars_asked": int(a.chars), "haps": a.haps, "txp_len": a.txp, "subs": a.subs}
    log = lambda *x: print(*x, file=sys.stderr, flush=True)
    rng = np.random.default_rng(11)
    ntx = int(a.chars // a.haps // (a.txp + 1))          # transcripts per haplotype ('$' separators count into the text)
    base = rng.integers(0, 4, ntx * a.txp, dtype=np.uint8)
    fa = os.path.join(a.dir, "pan.fa")
    t = time.time()
    LUT = np.frombuffer(b"ACGT", dtype=np.uint8)
    with open(fa, "wb") as f:
        for h in range(a.haps):
            seq = base.copy()
            m = rng.random(seq.size) < a.subs
            seq[m] = (seq[m] + rng.integers(1, 4, int(m.sum()), dtype=np.uint8)) & 3
            txt = LUT[seq].reshape(ntx, a.txp)
            rows = np.empty((ntx, a.txp + 1), dtype=np.uint8); rows[:, :-1] = txt; rows[:, -1] = 10
            for i0 in range(0, ntx, 4096):
                blk = rows[i0:i0 + 4096]
                f.write(b"".join(b">h%d_t%d\n" % (h, i0 + j) + blk[j].tobytes() for j in range(blk.shape[0])))
            if h == a.haps - 1:
                last = txt.copy()
    out["fasta_seconds"] = round(time.time() - t, 1)
    log("fasta: %d transcripts of %d, %.1f s" % (a.haps * ntx, a.txp, out["fasta_seconds"]))
    idx = os.path.join(a.dir, "idx")
    if a.force_big:
        os.environ["QM_FORCE_BIGSA"] = "1"
    t = time.time()
    ra.build_index(fa, idx, threads=a.threads, no_clip_poly_a=True)
    out["index_build_seconds"] = round(time.time() - t, 1)
    log("index: %.1f s" % out["index_build_seconds"])
    hdr = json.load(open(os.path.join(idx, "header.json")))["value0"]
    assert hdr["BigSA"] is True
    t = time.time()
    qi = ra.QuasiIndex(idx)
    out["open_seconds"] = round(time.time() - t, 1)
    out.update(text_len=qi.text_len, n_txps=qi.n_txps, n_keys=qi.n_keys, big_sa=qi.big_sa, beyond_int32=qi.text_len > 2**31 - 1)
    log("open: %.1f s, text %d, keys %d" % (out["open_seconds"], qi.text_len, qi.n_keys))
    # pairs: half from the last haplotype (text positions beyond 2^31 when the text is), half from the base; 1 % substitutions
    L, n = 100, a.pairs
    comp = np.array([3, 2, 1, 0], dtype=np.uint8)
    s1 = np.empty((n, L), dtype=np.uint8); s2 = np.empty((n, L), dtype=np.uint8)
    bt = base.reshape(ntx, a.txp)
    lcode = np.zeros(256, dtype=np.uint8); lcode[LUT] = np.arange(4, dtype=np.uint8)
    lastc = lcode[last]
    tx = rng.integers(0, ntx, n); st = rng.integers(0, a.txp - 300, n); fl = rng.integers(2 * L, 300, n)
    for i in range(n):
        src = lastc if i & 1 else bt
        frag = src[tx[i], st[i]:st[i] + fl[i]]
        s1[i] = frag[:L]; s2[i] = comp[frag[-L:][::-1]]
    for s in (s1, s2):
        m = rng.random(s.shape) < 0.01
        s[m] = (s[m] + rng.integers(1, 4, int(m.sum()), dtype=np.uint8)) & 3
    q1 = LUT[s1].reshape(-1); q2 = LUT[s2].reshape(-1)
    off = np.arange(n + 1, dtype=np.int64) * L
    if not a.no_gpu:
        t = time.time()
        mp = ra.QuasiMapper(qi, 0, debug=True)
        out["replica_seconds"] = round(time.time() - t, 1); out["device_bytes"] = mp.device_bytes
        log("replica: %.1f s, %.1f GB" % (out["replica_seconds"], mp.device_bytes / 1e9))
        gr = mp.map_pairs(q1, off, q2, off)
        t = time.time(); gr = mp.map_pairs(q1, off, q2, off); out["gpu_map_seconds"] = round(time.time() - t, 3)
        goffs, gints = mp.intervals(n)
        sel = mp.map_pairs(q1[: 20000 * L], off[:20001], q2[: 20000 * L], off[:20001], opts=ra.default_opts(sel_aln=1))
    t = time.time()
    ix = q5.load(idx)
    orc = oracle.Oracle(ix)
    out["oracle_load_seconds"] = round(time.time() - t, 1)
    assert ix.SA.dtype == np.int64 and orc.lib.qo_index_bytes() == 8
    t = time.time()
    res = orc.map_pairs(q1, off, q2, off, nthreads=a.threads, want_ints=True)
    out["oracle_map_seconds"] = round(time.time() - t, 1)
    b = res.ints[:, 0].astype(np.int64) & 0xffffffff
    out["oracle"] = {"hits": int(res.hit_offsets[-1]), "counters": res.counters, "interval_records": int(res.ints.shape[0]),
                     "interval_bounds_beyond_2^31": int((b > 2**31 - 1).sum()),
                     "suffixes_at_text_positions_beyond_2^31": int((ix.SA > 2**31 - 1).sum())}
    if not a.no_gpu:
        ok = bool(np.array_equal(res.hit_offsets, gr.hit_offsets) and res.hits.tobytes() == gr.hits.tobytes() and res.counters == gr.counters)
        iok = bool(np.array_equal(res.ints_offsets, goffs) and all(
            np.array_equal(res.ints[:, c], gints[nm].astype(np.int32)) for c, nm in ((0, "begin"), (1, "end"), (2, "len"), (3, "query_pos"), (5, "list"))))
        rs = orc.map_pairs(q1[: 20000 * L], off[:20001], q2[: 20000 * L], off[:20001], opts=oracle.default_opts(selAln=1), nthreads=a.threads)
        sok = bool(np.array_equal(rs.hit_offsets, sel.hit_offsets) and rs.hits.tobytes() == sel.hits.tobytes() and rs.counters == sel.counters)
        out["gpu"] = {"hits": int(gr.hit_offsets[-1]), "bit_identical_hits_and_counters": ok, "bit_identical_interval_records": iok,
                      "selective_alignment_20k_pairs_bit_identical": sok, "pairs": n}
        assert ok and iok and sok, out
    print(json.dumps(out))


if __name__ == "__main__":
    main()
