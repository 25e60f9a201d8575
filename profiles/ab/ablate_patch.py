#!/usr/bin/env python3
"""profiles/ab/ablate_patch.py <csrc dir>: early returns at phase boundaries of stage A, selected by -DQM_ABLATE=n, so that the
per-phase instruction counts can be read off PMC deltas between variants (there is no PC sampling / thread trace on the box):
  1 read staged, upper-cased (no collector)      2 + forward strand set up      3 + first-hit scan
  4 + forward get_sa_hits                         5 whole collector, no hits->mappings / write-out      6 + hits->mappings, empty list written      (unset: everything)
Results are wrong by construction; only the counters of these builds mean anything."""
import sys
p = sys.argv[1] + "/qm_mapper.inl"
s = open(p).read()
def ins(anchor, text, after=True):
    global s
    assert anchor in s, anchor
    s = s.replace(anchor, anchor + "\n" + text if after else text + "\n" + anchor, 1)
ins("  IntervalList fi, ri;\n  fi.lds = (QM_LDS(IntRec)*)M.ints[0]; ri.lds = (QM_LDS(IntRec)*)M.ints[1];",
    "#if defined(QM_ABLATE) && QM_ABLATE == 1\n  lds_dma_wait(); QM_LANES(l) { if (l == 0) { B.lcnt[read] = 0; B.loff[read] = 0; } } return;\n#endif")
ins("  setup_strand<NS>(ix, fwdStr, L, S, &M.planes[0][0][0], M.tab[0]);\n  S.dollar = hasDollar;",
    "#if defined(QM_ABLATE) && QM_ABLATE == 2\n  return S.P == 12345;\n#endif")
ins("  if (!found) return false;\n  const FT w0 = S.word_at(p0);",
    "#if defined(QM_ABLATE) && QM_ABLATE == 3\n  return w0 == 12345;\n#endif")
ins("  bool checkRC = useCoverageCheck ? (rcHit > 0) : (rcHit >= fwdHit);",
    "#if defined(QM_ABLATE) && QM_ABLATE == 4\n  return true;\n#endif")
ins("  if (F & QM_F_COLLECT) return;          // stage entry",
    "#if defined(QM_ABLATE) && QM_ABLATE == 5\n  lds_dma_wait(); QM_LANES(l) { if (l == 0) { B.lcnt[read] = (u32)((fi.n + ri.n) & 0); B.loff[read] = 0; } } return;\n#endif", after=False)
ins("    listSrc = bf.R;\n  }\n  QM_T(5);",
    "#if defined(QM_ABLATE) && QM_ABLATE == 6\n  n = 0;\n#endif")
open(p, "w").write(s)
