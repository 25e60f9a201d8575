import collections
import gzip

import numpy as np


def pack(reads):
    from rapmap_amd import pack_reads
    return pack_reads(reads)


def sam_groups_from_gz(path):
    """expected_*.noseq.sam.gz -> {qname: [lines]} (SEQ column already removed)"""
    g = collections.defaultdict(list)
    with gzip.open(path, "rt") as f:
        for l in f:
            g[l.split("\t", 1)[0]].append(l)
    return g


def strip_seq(sam_text):
    out = []
    for l in sam_text.splitlines(True):
        c = l.split("\t")
        out.append("\t".join(c[:9] + c[10:]))
    return out


def assert_hits_equal(a_off, a_hits, b_off, b_hits, what=""):
    assert np.array_equal(a_off, b_off), "%s: hit offsets differ (first at unit %d)" % (
        what, int(np.nonzero(a_off != b_off)[0][0]) - 1)
    if a_hits.tobytes() != b_hits.tobytes():
        bad = [i for i in range(len(a_off) - 1)
               if a_hits[a_off[i]:a_off[i + 1]].tobytes() != b_hits[b_off[i]:b_off[i + 1]].tobytes()]
        i = bad[0]
        raise AssertionError("%s: %d units differ; first unit %d:\n%s\nvs\n%s" % (
            what, len(bad), i, a_hits[a_off[i]:a_off[i + 1]], b_hits[b_off[i]:b_off[i + 1]]))
