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


def write_bgzf(path, data, block=60000, level=6, eof_marker=True):
    """`data` as a BGZF file (SAM specification 4.1; what `bgzip` writes): complete gzip members of at most 64 KiB, each with
    the 'BC' extra field that holds its compressed size - 1"""
    import struct
    import zlib
    with open(path, "wb") as f:
        chunks = [data[i:i + block] for i in range(0, len(data), block)] + ([b""] if eof_marker else [])
        for c in chunks:
            co = zlib.compressobj(level, zlib.DEFLATED, -15)
            body = co.compress(c) + co.flush()
            bsize = 12 + 6 + len(body) + 8
            f.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1))
            f.write(body + struct.pack("<II", zlib.crc32(c) & 0xffffffff, len(c) & 0xffffffff))
