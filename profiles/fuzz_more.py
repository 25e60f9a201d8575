#!/usr/bin/env python3
"""Extra fuzzing of the HIP path against the oracle beyond what the -m gpu suite runs each time: more seeds, more read-length
classes, more option sets.  python profiles/fuzz_more.py [n_pairs] [seeds] [first seed]  (GPU box; builds a small synthetic index)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch; torch.cuda.init()
import rapmap_amd as ra
from rapmap_amd import synth
from oracle import oracle, q5
from util import pack, assert_hits_equal
import test_gpu_parity as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
fuzz_n = float(os.environ.get("FUZZ_N", "0"))
oracle.build()
d = tempfile.mkdtemp(prefix="fuzz", dir="/dev/shm")
names, txps = synth.make_transcriptome(600, seed=11)
fa = os.path.join(d, "t.fa"); synth.write_fasta(fa, names, txps)
bad = 0
for ph in (False, True):
    idx = os.path.join(d, "idx_ph" if ph else "idx"); ra.build_index(fa, idx, threads=16, perfect_hash=ph)
    orc = oracle.Oracle(q5.load(idx)); qi = ra.QuasiIndex(idx)
    text, offsets = qi.arrays()
    for compact in ((False, True) if ph else (False,)):
        mp = ra.QuasiMapper(qi, 0, ph_compact=compact)
        for seed in range(seed0, seed0 + seeds):
            for max_len in (100, 128, 129, 150, 192, 250, 256, 257, 400):
                r1, r2 = T._fuzz_reads(np.asarray(text), np.asarray(offsets, dtype=np.int64), n if max_len <= 256 else n // 4, 1000 * seed + max_len, max_len)
                if fuzz_n > 0:                                   # FUZZ_N=rate: N's (upper and lower case) sprinkled over the reads on top of the generator's own
                    rngn = np.random.default_rng(77 + 1000 * seed + max_len)
                    def sprinkle(rs):
                        out = []
                        for r in rs:
                            a = np.frombuffer(r, np.uint8).copy()
                            if a.size:
                                w = rngn.random(a.size) < fuzz_n
                                a[w] = rngn.choice(np.frombuffer(b"NNNn", np.uint8), int(w.sum()))
                            out.append(a.tobytes())
                        return out
                    r1, r2 = sprinkle(r1), sprinkle(r2)
                q1, o1 = pack(r1); q2, o2 = pack(r2)
                sets = [({}, {}), ({"sensitive": 0}, {"sensitive": 0}), ({"fuzzy": 1}, {"fuzzy": 1}), ({"strictCheck": 0}, {"strict_check": 0}),
                        ({"maxNumHits": 2, "noOrphans": 1}, {"max_num_hits": 2, "no_orphans": 1}), ({"quasiCov": 0.8}, {"quasi_cov": 0.8}),
                        ({"noDovetail": 1, "maxNumHits": 3}, {"no_dovetail": 1, "max_num_hits": 3})]       # (round 6: the pair kernel merges pairs itself: its --noDovetail / maxNumHits rules)
                if max_len <= 250:
                    sets += [({"selAln": 1}, {"sel_aln": 1}), ({"selAln": 1, "consensusSlack": 0.35, "dpBandwidth": 40}, {"sel_aln": 1, "consensus_slack": 0.35, "dp_bandwidth": 40}),
                             ({"selAln": 1, "quasiCov": 0.7}, {"sel_aln": 1, "quasi_cov": 0.7}), ({"selAln": 1, "strictCheck": 0, "maxMMPExtension": 3}, {"sel_aln": 1, "strict_check": 0, "max_mmp_extension": 3})]      # (round 6: the collector's coverage sums and runs under another stride)
                for oo, go in sets:
                    res = orc.map_pairs(q1, o1, q2, o2, opts=oracle.default_opts(**oo), nthreads=32)
                    gr = mp.map_pairs(q1, o1, q2, o2, opts=ra.default_opts(**go))
                    try:
                        assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "fuzz")
                        assert res.counters == gr.counters
                    except AssertionError as e:
                        bad += 1; print("MISMATCH ph=%s compact=%s seed=%d len<=%d opts=%s: %s" % (ph, compact, seed, max_len, oo, str(e)[:200]), flush=True)
            print("ph=%s compact=%s seed %d done" % (ph, compact, seed), flush=True)
        if not ph:
            # round 5: -s on reads whose edits sit where the gapless path and the one-gap-run paths differ (two substitutions, indels a few
            # characters from either end), under several score schemes: every answer sel_side_score gives without ksw2 against the oracle's ksw2
            import test_emu_parity as E
            for seed in range(seed0, seed0 + seeds):
                r1, r2 = E._edited_pairs(txps, 4 * n, 777 + seed)
                q1, o1 = pack(r1); q2, o2 = pack(r2)
                for oo, go in [({}, {}), ({"gapOpen": 5, "gapExtend": 3}, {"gap_open": 5, "gap_extend": 3}), ({"matchScore": 1, "mismatchPenalty": -1, "gapOpen": 1, "gapExtend": 1},
                               {"match_score": 1, "mismatch_penalty": -1, "gap_open": 1, "gap_extend": 1}), ({"dpBandwidth": 5}, {"dp_bandwidth": 5}), ({"dpBandwidth": -1, "gapOpen": 6, "gapExtend": 1},
                               {"dp_bandwidth": -1, "gap_open": 6, "gap_extend": 1}), ({"hardFilter": 1, "minScoreFraction": 0.3}, {"hard_filter": 1, "min_score_fraction": 0.3})]:
                    res = orc.map_pairs(q1, o1, q2, o2, opts=oracle.default_opts(selAln=1, **oo), nthreads=32)
                    gr = mp.map_pairs(q1, o1, q2, o2, opts=ra.default_opts(sel_aln=1, **go))
                    try:
                        assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "fuzz -s edited")
                        assert res.counters == gr.counters
                    except AssertionError as e:
                        bad += 1; print("MISMATCH -s edited seed=%d opts=%s: %s" % (seed, oo, str(e)[:200]), flush=True)
                print("-s edited reads seed %d done: %d pairs, %d questions, %d ksw2 alignments in the last run" % (seed, len(r1), mp.stat(6), mp.stat(7)), flush=True)
        npass_last = mp.stat(15)
        mp.close()
print("fuzz_more: %d mismatching runs (FUZZ_N=%g, QM_NPASS_MIN=%s; the last call's N-aware pass mapped %d reads)" % (bad, fuzz_n, os.environ.get("QM_NPASS_MIN", "default"), npass_last))
sys.exit(1 if bad else 0)
