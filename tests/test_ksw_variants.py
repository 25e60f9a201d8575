"""The ksw2 extension kernel of the device source (rapmap_amd/csrc/qm_sel.inl: sel_ksw_extz2_rows, four alignments per
wavefront, one kernel for every --dpBandwidth -- run here through the lane emulation) against the oracle's restatement of
ksw_extz2_sse41 (oracle/qm_oracle.cpp: kswExtz2, itself checked against the reference's own ksw2 sources compiled into
oracle/_ref and against the reference's `-s` SAM fixtures), on random and adversarial inputs:
every target length 1..160 (all residues mod 16: the SSE vectors' padding lanes and the band's last rounds differ),
indels, N's, short and long queries, several bands and scoring schemes."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(HERE, "emu"))
from oracle import oracle  # noqa: E402
import emu  # noqa: E402


def _mutate(rng, q, tlen, sub=0.05, indel=0.02, nfrac=0.01):
    out = []
    i = 0
    while len(out) < tlen:
        r = rng.random()
        if r < indel:            # insertion in the target
            out.append(rng.integers(0, 4))
        elif r < 2 * indel:      # deletion
            i += 1
        else:
            c = q[i % len(q)]
            if rng.random() < sub:
                c = rng.integers(0, 4)
            if rng.random() < nfrac:
                c = 4
            out.append(c); i += 1
    return np.array(out[:tlen], dtype=np.uint8)


def _cases(seed, n):
    rng = np.random.default_rng(seed)
    for it in range(n):
        qlen = int(rng.integers(1, 141))
        tlen = int(rng.integers(1, 161)) if it % 3 else min(160, qlen + 20)
        q = rng.integers(0, 4, qlen).astype(np.uint8)
        if rng.random() < 0.1:
            q[rng.integers(0, qlen)] = 4
        t = _mutate(rng, q, tlen) if rng.random() < 0.85 else rng.integers(0, 5, tlen).astype(np.uint8)
        yield q, t


# (match, mismatch, gap open, gap extend, band): bands <= 15 run on the register edition of the row kernel ("32"), <= 33 on the
# 64-slot ring, <= 97 on 128 slots, everything else (and the "whole matrix" band -1) on 1024
SCHEMES = [(2, -4, 4, 2, 15), (2, -4, 4, 2, 5), (1, -1, 1, 1, 15), (2, -6, 5, 3, 33), (2, -4, 4, 2, 0), (4, -4, 6, 2, 20),
           (2, -4, 4, 2, 1), (2, -6, 5, 3, 8), (1, 0, 25, 25, 15), (2, -4, 4, 2, 14), (2, -4, 4, 2, 16),
           (2, -4, 4, 2, 34), (2, -4, 4, 2, 40), (2, -6, 5, 3, 64), (1, -1, 1, 1, 97), (2, -4, 4, 2, 98), (2, -4, 4, 2, 150),
           (2, -4, 4, 2, 1000), (2, -4, 4, 2, -1)]


def _rows(el, grp, a, b, q_, e_, w, ring=-1):
    ql = (C.c_int * 4)(*[len(g[0]) for g in grp] + [0] * (4 - len(grp)))
    tl = (C.c_int * 4)(*[len(g[1]) for g in grp] + [0] * (4 - len(grp)))
    dummy = np.zeros(1, dtype=np.uint8)
    qp = (C.c_void_p * 4)(*[g[0].ctypes.data for g in grp] + [dummy.ctypes.data] * (4 - len(grp)))
    tp = (C.c_void_p * 4)(*[g[1].ctypes.data for g in grp] + [dummy.ctypes.data] * (4 - len(grp)))
    out = (C.c_int * 4)()
    el.qe_ksw_rows(ql, qp, tl, tp, a, b, q_, e_, w, out, ring)
    return list(out)


@pytest.mark.parametrize("scheme", SCHEMES)
def test_ksw_rows_kernel_four_at_a_time(scheme):
    """the 16-lane-row kernel (four alignments of different shapes per wavefront, idle rows included) against the oracle"""
    a, b, q_, e_, w = scheme
    ol = oracle._lib(); el = emu._lib()
    ol.qo_ksw_extz2.restype = C.c_int
    cases = list(_cases(4321 + w, 400))
    rng = np.random.default_rng(5)
    bad = []
    for i in range(0, len(cases), 4):
        grp = cases[i:i + 4]
        if rng.random() < 0.2:
            grp = grp[:int(rng.integers(1, 4))]            # a partly filled wavefront
        out = _rows(el, grp, a, b, q_, e_, w)
        for k, (qq, tt) in enumerate(grp):
            ref = ol.qo_ksw_extz2(len(qq), qq.ctypes.data_as(C.c_void_p), len(tt), tt.ctypes.data_as(C.c_void_p), a, b, q_, e_, w)
            if out[k] != ref:
                bad.append((k, len(qq), len(tt), ref, out[k]))
    assert not bad, "%d mismatches, first: %r" % (len(bad), bad[:5])


@pytest.mark.parametrize("ring", [32, 64, 128, 1024])
def test_ksw_rows_every_ring_gives_the_same_scores(ring):
    """a band that fits the smallest ring must score the same on the larger ones (the ring is storage, not arithmetic); 32: the
    register edition, bands up to 15 only"""
    ol = oracle._lib(); el = emu._lib()
    ol.qo_ksw_extz2.restype = C.c_int
    cases = list(_cases(99, 200))
    for w in ((15, 7) if ring == 32 else (15, 33)):
        for i in range(0, len(cases), 4):
            grp = cases[i:i + 4]
            out = _rows(el, grp, 2, -4, 4, 2, w, ring)
            for k, (qq, tt) in enumerate(grp):
                ref = ol.qo_ksw_extz2(len(qq), qq.ctypes.data_as(C.c_void_p), len(tt), tt.ctypes.data_as(C.c_void_p), 2, -4, 4, 2, w)
                assert out[k] == ref, (ring, w, len(qq), len(tt), ref, out[k])


@pytest.mark.parametrize("qmax", [256, 512])
def test_ksw_rows_longest_alignments(qmax):
    """queries of up to 512 bases against targets 20 longer (the longest the -s path produces), narrow to full band"""
    ol = oracle._lib(); el = emu._lib()
    ol.qo_ksw_extz2.restype = C.c_int
    rng = np.random.default_rng(31 + qmax)
    for w in (15, 33, 60, 97, 200, -1):
        grp = []
        for _ in range(4):
            q = rng.integers(0, 4, int(rng.integers(qmax - 26, qmax + 1))).astype(np.uint8)
            t = _mutate(rng, q, len(q) + 20, sub=0.03, indel=0.01)
            grp.append((q, t))
        out = _rows(el, grp, 2, -4, 4, 2, w)
        for k, (qq, tt) in enumerate(grp):
            ref = ol.qo_ksw_extz2(len(qq), qq.ctypes.data_as(C.c_void_p), len(tt), tt.ctypes.data_as(C.c_void_p), 2, -4, 4, 2, w)
            assert out[k] == ref, (w, len(qq), len(tt), ref, out[k])


@pytest.mark.parametrize("w", [15, 40, 120])
def test_ksw_rows_every_target_length(w):
    """query 100, perfect and 1-error targets of every length 1..160: the rounds where the band has shrunk to its last cells
    sit at a different place of the 16-byte vectors for every residue of tlen mod 16"""
    ol = oracle._lib(); el = emu._lib()
    ol.qo_ksw_extz2.restype = C.c_int
    rng = np.random.default_rng(9)
    for t0 in range(1, 161, 4):
        grp = []
        for tlen in range(t0, t0 + 4):
            q = rng.integers(0, 4, 100).astype(np.uint8)
            t = np.resize(q, tlen).copy()
            if tlen > 3:
                t[tlen // 2] ^= 1
            grp.append((q, t))
        out = _rows(el, grp, 2, -4, 4, 2, w)
        for k in range(4):
            qq, tt = grp[k]
            ref = ol.qo_ksw_extz2(100, qq.ctypes.data_as(C.c_void_p), len(tt), tt.ctypes.data_as(C.c_void_p), 2, -4, 4, 2, w)
            assert out[k] == ref, (w, t0 + k, ref, out[k])


@pytest.mark.parametrize("scheme", [(2, -4, 4, 2, 15), (2, -6, 5, 3, 8), (1, -1, 1, 1, 0), (2, -4, 4, 2, 5), (1, 0, 25, 25, 15)])
def test_ksw_rows_same_shape_wavefronts(scheme):
    """four (or fewer: idle rows) alignments of ONE shape per wavefront -- the register edition then keeps the band's geometry on
    the scalar unit (sel_ksw_extz2_rows_reg<.., UNI = true>): every query length class and target length residue, indels, N's"""
    a, b, q_, e_, w = scheme
    ol = oracle._lib(); el = emu._lib()
    ol.qo_ksw_extz2.restype = C.c_int
    rng = np.random.default_rng(77 + w)
    bad = []
    shapes = [(100, 120), (100, 100), (100, 87), (1, 1), (1, 21), (31, 51), (75, 95), (140, 160), (50, 1), (16, 16), (17, 33), (128, 148)]
    shapes += [(int(rng.integers(1, 141)), int(rng.integers(1, 161))) for _ in range(60)]
    for qlen, tlen in shapes:
        for n in (4, 4, int(rng.integers(1, 4))):
            grp = []
            for _ in range(n):
                q = rng.integers(0, 4, qlen).astype(np.uint8)
                if rng.random() < 0.2:
                    q[rng.integers(0, qlen)] = 4
                t = _mutate(rng, q, tlen) if rng.random() < 0.85 else rng.integers(0, 5, tlen).astype(np.uint8)
                grp.append((q, t))
            out = _rows(el, grp, a, b, q_, e_, w)
            for k, (qq, tt) in enumerate(grp):
                ref = ol.qo_ksw_extz2(len(qq), qq.ctypes.data_as(C.c_void_p), len(tt), tt.ctypes.data_as(C.c_void_p), a, b, q_, e_, w)
                if out[k] != ref:
                    bad.append((k, n, len(qq), len(tt), ref, out[k]))
    assert not bad, "%d mismatches, first: %r" % (len(bad), bad[:5])


@pytest.mark.parametrize("scheme", [(2, -4, 4, 2, 15), (2, -6, 5, 3, 8), (1, -1, 1, 1, 0), (2, -4, 4, 2, 5)])
def test_ksw_rows_eight_per_wavefront(scheme):
    """eight alignments of one shape as two sets of state through the same rounds (sel_ksw_extz2_rows_reg<.., UNI, 2>, what
    qm_sel_align2_kernel runs when a wavefront's eight tasks share their lengths), 5 .. 8 of them real"""
    a, b, q_, e_, w = scheme
    ol = oracle._lib(); el = emu._lib()
    ol.qo_ksw_extz2.restype = C.c_int
    rng = np.random.default_rng(177 + w)
    bad = []
    shapes = [(100, 120), (100, 100), (100, 87), (1, 1), (31, 51), (75, 95), (140, 160), (16, 16), (17, 33), (128, 148)]
    shapes += [(int(rng.integers(1, 141)), int(rng.integers(1, 161))) for _ in range(40)]
    for qlen, tlen in shapes:
        for n in (8, int(rng.integers(5, 8))):
            grp = []
            for _ in range(n):
                q = rng.integers(0, 4, qlen).astype(np.uint8)
                if rng.random() < 0.2:
                    q[rng.integers(0, qlen)] = 4
                t = _mutate(rng, q, tlen) if rng.random() < 0.85 else rng.integers(0, 5, tlen).astype(np.uint8)
                grp.append((q, t))
            dummy = np.zeros(1, dtype=np.uint8)
            qp = (C.c_void_p * 8)(*[g[0].ctypes.data for g in grp] + [dummy.ctypes.data] * (8 - n))
            tp = (C.c_void_p * 8)(*[g[1].ctypes.data for g in grp] + [dummy.ctypes.data] * (8 - n))
            out = (C.c_int * 8)()
            el.qe_ksw_rows8(n, qlen, qp, tlen, tp, a, b, q_, e_, w, out)
            for k, (qq, tt) in enumerate(grp):
                ref = ol.qo_ksw_extz2(len(qq), qq.ctypes.data_as(C.c_void_p), len(tt), tt.ctypes.data_as(C.c_void_p), a, b, q_, e_, w)
                if out[k] != ref:
                    bad.append((k, n, qlen, tlen, ref, out[k]))
    assert not bad, "%d mismatches, first: %r" % (len(bad), bad[:5])


@pytest.mark.parametrize("scheme", [(2, -4, 4, 2, 15), (2, -4, 4, 2, 1), (2, -4, 4, 2, 5), (1, -1, 1, 1, 15), (2, -6, 5, 3, 33), (4, -4, 6, 2, 20),
                                    (2, -4, 4, 2, 98), (2, -4, 4, 2, -1), (3, -2, 2, 1, 15), (1, 0, 25, 25, 15)])
def test_gapless_path_wins_when_it_loses_no_more_than_one_gap(scheme):
    """sel_side_score's first rule (rapmap_amd/csrc/qm_sel.inl): with the target at least as long as the query, an extension alignment
    whose gapless path loses at most q + e against 'every query character at its best' scores exactly what that path scores.
    Held against the oracle's ksw_extz2 (itself pinned to the reference's kernel compiled in place) on queries with zero, one or
    two differences, N's on either side, indels right behind the start, every length class and band (but --dpBandwidth 0, whose
    odd anti-diagonals are empty: the kernel stops at the second one, and the shortcut stays away from it)."""
    a, b, q_, e_, w = scheme
    ol = oracle._lib()
    ol.qo_ksw_extz2.restype = C.c_int
    rng = np.random.default_rng(1234 + w)
    hits = 0
    for it in range(4000):
        qlen = int(rng.integers(1, 141))
        tlen = qlen + int(rng.integers(0, 25))
        q = rng.integers(0, 4, qlen).astype(np.uint8)
        t = np.concatenate([q, rng.integers(0, 4, tlen - qlen).astype(np.uint8)])
        for _ in range(int(rng.integers(0, 3))):              # substitutions in the target
            t[rng.integers(0, qlen)] = rng.integers(0, 4)
        if rng.random() < 0.15: q[rng.integers(0, qlen)] = 4  # an N in the query
        if rng.random() < 0.15: t[rng.integers(0, tlen)] = 4  # ... in the target
        if rng.random() < 0.1 and qlen > 4:                   # a deletion: the target continues one character further on
            p = int(rng.integers(0, qlen - 1)); t = np.concatenate([t[:p], t[p + 1:], rng.integers(0, 4, 1).astype(np.uint8)])
        aa, bb = abs(a), -abs(b)
        sc = np.where((q < 4) & (t[:qlen] < 4), np.where(q == t[:qlen], aa, bb), 0)
        smax = int(np.where(q < 4, aa, 0).sum()); U = int(sc.sum())
        if smax - U <= q_ + e_:
            ref = ol.qo_ksw_extz2(qlen, q.ctypes.data_as(C.c_void_p), len(t), t.ctypes.data_as(C.c_void_p), a, b, q_, e_, w)
            assert ref == U, (scheme, qlen, tlen, smax, U, ref)
            hits += 1
    assert hits > 500


def _known_answer_rule(q, t, a, b, q_, e_, w):
    """returns score or None (ask ksw2)"""
    qlen, tlen = len(q), len(t)
    aa, bb = abs(a), -abs(b)
    M = aa - bb; qe = q_ + e_
    if not (w != 0 and tlen >= qlen and qlen > 0 and aa >= 1 and q_ >= 0 and e_ >= 1 and aa - bb + qe <= 96): return None
    sc = np.where((q < 4) & (t[:qlen] < 4), np.where(q == t[:qlen], aa, bb), 0)
    smax = int(np.where(q < 4, aa, 0).sum()); U = int(sc.sum()); loss = smax - U
    if loss <= qe: return U
    # extension
    if M > qe or loss != 2 * M: return None
    Ld = 0
    while q_ + (Ld + 1) * e_ < 2 * M: Ld += 1
    Li = 0
    while q_ + (Li + 1) * (e_ + aa) < 2 * M: Li += 1
    if Ld > 3 or Li > 2: return None
    if not (w < 0 or w >= 8): return None
    if tlen < qlen + Ld + 1: return None
    if (q >= 4).any() or (t[:qlen + Ld] >= 4).any(): return None
    mm = np.nonzero(q != t[:qlen])[0]
    assert len(mm) == 2
    m1 = int(mm[0])
    best = 2 * M
    for L in range(1, Ld + 1):
        d = np.nonzero(q != t[L:L + qlen])[0]
        hm = int(d[-1]) if len(d) else -1
        if hm <= m1 - 1: best = min(best, q_ + L * e_)
    for L in range(1, Li + 1):
        idx = np.arange(L, qlen)
        d = idx[q[L:] != t[:qlen - L]]
        hm = int(d[-1]) if len(d) else L - 1
        if hm <= m1 + L - 1: best = min(best, q_ + L * (e_ + aa))
    return smax - best


@pytest.mark.parametrize("scheme", [(2, -4, 5, 3, 15), (2, -4, 4, 2, 15), (2, -4, 4, 2, 8), (1, -1, 1, 1, 15), (2, -6, 5, 3, 33), (4, -4, 6, 2, 20),
                                    (2, -4, 4, 2, -1), (3, -2, 2, 1, 15), (2, -4, 5, 3, 9), (1, -3, 2, 1, 15), (2, -4, 6, 1, 15)])
def test_two_mismatches_lose_to_at_most_one_gap_run(scheme):
    """sel_side_score's second rule (rapmap_amd/csrc/qm_sel.inl), restated above in numpy: an alignment whose gapless path has exactly
    two mismatches and no N scores the best of that path and the one-gap-run paths without a mismatch -- found by looking at the last
    mismatch of the diagonals next to the main one.  Held against the oracle's ksw_extz2 on queries from low-complexity and periodic
    targets (where the neighbouring diagonals do match), with the substitutions at the ends, true indels, N's."""
    a, b, q_, e_, w = scheme
    ol = oracle._lib()
    ol.qo_ksw_extz2.restype = C.c_int
    rng = np.random.default_rng(7 + w + q_)
    tot = ext = gapwin = 0
    for it in range(6000):
        qlen = int(rng.integers(3, 141))
        tlen = qlen + int(rng.integers(0, 25))
        alpha = int(rng.choice([1, 2, 2, 4, 4, 4]))
        if rng.random() < 0.3:
            per = int(rng.integers(1, 4)); unit = rng.integers(0, 4, per)
            base = np.tile(unit, (tlen + 8) // per + 1)[:tlen + 8].astype(np.uint8)
        else:
            base = rng.integers(0, alpha, tlen + 8).astype(np.uint8)
        t = base[:tlen].copy(); q = base[:qlen].copy()
        nsub = int(rng.choice([2, 2, 2, 1, 3]))
        pos = rng.choice(qlen, size=min(nsub, qlen), replace=False)
        if rng.random() < 0.4 and qlen > 6: pos = np.array([qlen - 1 - int(rng.integers(0, 3)), qlen - 1 - int(rng.integers(3, 6))])[:nsub]
        if rng.random() < 0.2 and qlen > 6: pos = np.array([int(rng.integers(0, 3)), int(rng.integers(3, 6))])[:nsub]
        for p in pos: q[p] = (q[p] + 1 + rng.integers(0, 3)) % 4
        if rng.random() < 0.15 and qlen > 5:
            p = int(rng.integers(0, qlen - 1))
            if rng.random() < 0.5: q = np.concatenate([q[:p], q[p + 1:], base[qlen:qlen + 1]])
            else: q = np.concatenate([q[:p], rng.integers(0, 4, 1).astype(np.uint8), q[p:-1]])
        if rng.random() < 0.03: q[rng.integers(0, qlen)] = 4
        if rng.random() < 0.03: t[rng.integers(0, tlen)] = 4
        r = _known_answer_rule(q, t, a, b, q_, e_, w)
        if r is None: continue
        ref = ol.qo_ksw_extz2(qlen, q.ctypes.data_as(C.c_void_p), len(t), t.ctypes.data_as(C.c_void_p), a, b, q_, e_, w)
        assert r == ref, (scheme, qlen, tlen, r, ref, q.tolist(), t.tolist())
        tot += 1
        aa, bb = abs(a), -abs(b)
        U = int(np.where((q < 4) & (t[:qlen] < 4), np.where(q == t[:qlen], aa, bb), 0).sum())
        if int(np.where(q < 4, aa, 0).sum()) - U > q_ + e_:
            ext += 1; gapwin += r != U
    M = abs(a) + abs(b)
    if M <= q_ + e_: assert tot > 1000
    if M <= q_ + e_ and (w < 0 or w >= 8) and q_ + 4 * e_ >= 2 * M and q_ + 3 * (e_ + abs(a)) >= 2 * M:   # (the rule's own limits)
        assert ext > 300 and gapwin > 0, (tot, ext, gapwin)


NEG = -(1 << 28)

def _strip_dp(q, t, a, b, go, ge, half=7):
    """extension alignment from (0,0), affine gaps go + L*ge, score-only, max(mqe, mte); diagonals -half .. half+1"""
    qlen, tlen = len(q), len(t)
    aa, bb = abs(a), -abs(b)
    W = 2 * half + 2
    Hp = [NEG] * W; Fp = [NEG] * W
    for c in range(W):
        j = c - (half + 1)
        if j == -1: Hp[c] = 0
        elif 0 <= j < tlen: Hp[c] = -(go + ge * (j + 1))
    mqe = NEG; mte = NEG
    for i in range(qlen):
        Ht = [NEG] * W; F = [NEG] * W
        for c in range(W):
            j = i + c - half
            if j == -1:
                Ht[c] = -(go + ge * (i + 1)); continue
            if j < 0 or j >= tlen: continue
            s = (aa if q[i] == t[j] else bb) if (q[i] < 4 and t[j] < 4) else 0
            M = Hp[c] + s if Hp[c] > NEG else NEG
            f = NEG
            if c + 1 < W:
                if Hp[c + 1] > NEG: f = Hp[c + 1] - go - ge
                if Fp[c + 1] > NEG: f = max(f, Fp[c + 1] - ge)
            F[c] = f
            Ht[c] = max(M, f)
        H = [NEG] * W
        run = NEG
        for c in range(W):
            j = i + c - half
            E = run - go - ge * c if run > NEG else NEG
            if j == -1: H[c] = Ht[c]
            elif 0 <= j < tlen: H[c] = max(Ht[c], E)
            if Ht[c] > NEG: run = max(run, Ht[c] + ge * c)
            if 0 <= j < tlen:
                if i == qlen - 1: mqe = max(mqe, H[c])
                if j == tlen - 1: mte = max(mte, H[c])
        Hp, Fp = H, F
    return max(mqe, mte)


@pytest.mark.parametrize("scheme", [(2, -4, 4, 2, 15), (2, -4, 5, 3, 15), (1, -1, 1, 1, 15), (2, -6, 5, 3, 33), (2, -4, 4, 2, -1), (3, -2, 2, 1, 15), (2, -4, 6, 1, 15)])
def test_strip_dp_equals_ksw2_when_the_gapless_path_is_close(scheme):
    """sel_tasks_strip (rapmap_amd/csrc/qm_sel.inl), restated above: the extension alignment as a plain affine-gap recurrence over the
    diagonals -7 .. +8.  When the gapless path loses no more than q + 7 e no path that scores as much leaves that strip, so the strip's
    maximum over the query's last row and the target's last column is ksw_extz2's score -- held against the oracle's kernel on
    periodic / low-complexity targets, substitutions, true indels, N's and targets that end right behind the query."""
    a, b, go, ge, w = scheme
    ol = oracle._lib()
    ol.qo_ksw_extz2.restype = C.c_int
    rng = np.random.default_rng(3 + go + w)
    used = 0
    for it in range(2500):
        qlen = int(rng.integers(3, 141)); tlen = qlen + int(rng.integers(0, 25))
        alpha = int(rng.choice([1, 2, 4, 4, 4]))
        if rng.random() < 0.3:
            per = int(rng.integers(1, 4)); unit = rng.integers(0, 4, per); base = np.tile(unit, (tlen + 8) // per + 1)[:tlen + 8].astype(np.uint8)
        else:
            base = rng.integers(0, alpha, tlen + 8).astype(np.uint8)
        t = base[:tlen].copy(); q = base[:qlen].copy()
        for p in rng.choice(qlen, size=min(int(rng.integers(0, 5)), qlen), replace=False): q[p] = (q[p] + 1 + rng.integers(0, 3)) % 4
        if rng.random() < 0.3 and qlen > 5:
            p = int(rng.integers(0, qlen - 1))
            if rng.random() < 0.5: q = np.concatenate([q[:p], q[p + 1:], base[qlen:qlen + 1]])
            else: q = np.concatenate([q[:p], rng.integers(0, 4, 1).astype(np.uint8), q[p:-1]])
        if rng.random() < 0.1: q[rng.integers(0, qlen)] = 4
        if rng.random() < 0.1: t[rng.integers(0, tlen)] = 4
        aa, bb = abs(a), -abs(b)
        sc = np.where((q < 4) & (t[:qlen] < 4), np.where(q == t[:qlen], aa, bb), 0)
        loss = int(np.where(q < 4, aa, 0).sum()) - int(sc.sum())
        if not (tlen >= qlen and loss <= go + 7 * ge and (w < 0 or w >= 9) and aa - bb + go + ge <= 96): continue
        used += 1
        r = _strip_dp(q.tolist(), t.tolist(), a, b, go, ge)
        ref = ol.qo_ksw_extz2(qlen, q.ctypes.data_as(C.c_void_p), len(t), t.ctypes.data_as(C.c_void_p), a, b, go, ge, w)
        assert r == ref, (scheme, qlen, tlen, loss, r, ref, q.tolist(), t.tolist())
    assert used > 500
