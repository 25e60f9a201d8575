"""The library's native read ingest (qm_reader_*) and SAM writer (qm_sam_*) against the Python reader /
formatter and the reference digests.  No GPU: hit sets come from the oracle (the checker), the code under test
is host code of libqmap_mi355.so."""
import gzip
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLD, load_oracle
from util import pack


def _batches(path1, path2, chunk, threads=4):
    import rapmap_amd as ra
    rd = ra.FastxReader(path1, path2, threads=threads)
    out = []
    for b in rd.chunks(chunk):
        c = ra.ReadBatch(); c.n = b.n
        for k in ("seq1", "off1", "names1", "name_off1", "seq2", "off2", "names2", "name_off2"):
            if hasattr(b, k):
                setattr(c, k, np.array(getattr(b, k), copy=True))
        out.append(c)
    rd.close()
    return out


def _join(batches, key, okey):
    seqs = []
    for b in batches:
        a, o = getattr(b, key), getattr(b, okey)
        seqs += [a[o[i]:o[i + 1]].tobytes() for i in range(b.n)]
    return seqs


def test_reader_matches_python_reader(sample_data, tmp_path):
    import samfmt as sam
    p1 = os.path.join(GOLD, "sample_data", "reads_1.fastq.gz"); p2 = os.path.join(GOLD, "sample_data", "reads_2.fastq.gz")
    if not os.path.exists(p1):
        p1 = sample_data["paths"][0]; p2 = sample_data["paths"][1]
    n1, s1 = sam.read_fastq(p1); n2, s2 = sam.read_fastq(p2)
    # gzip input, odd chunk size
    bs = _batches(p1, p2, 777)
    assert sum(b.n for b in bs) == len(s1)
    assert _join(bs, "seq1", "off1") == s1 and _join(bs, "seq2", "off2") == s2
    assert [x.decode() for x in _join(bs, "names1", "name_off1")] == n1
    # plain text, large enough to be split across parser threads; also CRLF and a trailing record without newline
    plain1 = str(tmp_path / "r1.fq"); plain2 = str(tmp_path / "r2.fq")
    reps = 40
    with open(plain1, "wb") as f1, open(plain2, "wb") as f2:
        for r in range(reps):
            for i in range(len(s1)):
                f1.write(b"@%s\n%s\n+\n%s\n" % (n1[i].encode(), s1[i], b"@" * len(s1[i])))   # '@' qualities on purpose
                f2.write(b"@%s\r\n%s\r\n+\r\n%s\r\n" % (n2[i].encode(), s2[i], b"I" * len(s2[i])))
        f1.write(b"@last/1\nACGT\n+\nIIII")
        f2.write(b"@last/2\nTTGCA\n+\nIIIII")
    bs = _batches(plain1, plain2, 100000, threads=8)
    got1 = _join(bs, "seq1", "off1"); got2 = _join(bs, "seq2", "off2")
    assert got1 == s1 * reps + [b"ACGT"] and got2 == s2 * reps + [b"TTGCA"]
    nm = _join(bs, "names2", "name_off2")
    assert nm[-1] == b"last/2" and nm[0].decode() == n2[0]


def test_reader_fasta_and_errors(tmp_path):
    import rapmap_amd as ra
    fa = str(tmp_path / "x.fa")
    with open(fa, "w") as f:
        f.write(">a desc\nACGT\nAC\n>b\nGG\n\n>c\nT")
    (b,) = _batches(fa, None, 10)
    assert _join([b], "seq1", "off1") == [b"ACGTAC", b"GG", b"T"]
    assert _join([b], "names1", "name_off1") == [b"a desc", b"b", b"c"]
    # gzip'd multi-line FASTA, many records, tiny chunks (exercises the carry-over between decompressed blocks)
    import random
    rnd = random.Random(5)
    recs = [("s%d some text" % i, "".join(rnd.choice("ACGT") for _ in range(rnd.randint(1, 300)))) for i in range(3000)]
    gz = str(tmp_path / "y.fa.gz")
    with gzip.open(gz, "wt") as f:
        for nm, sq in recs:
            f.write(">%s\n" % nm)
            for j in range(0, len(sq), 60):
                f.write(sq[j:j + 60] + "\n")
    bs = _batches(gz, None, 257)
    assert _join(bs, "seq1", "off1") == [sq.encode() for _, sq in recs]
    assert _join(bs, "names1", "name_off1") == [nm.encode() for nm, _ in recs]
    # the same records 12x in a plain file: large enough for several parser threads (joined sequences per thread)
    big = str(tmp_path / "big.fa")
    with open(big, "w") as f:
        for r in range(12):
            for nm, sq in recs:
                f.write(">%s\n" % nm)
                for j in range(0, len(sq), 60):
                    f.write(sq[j:j + 60] + "\n")
    bs = _batches(big, None, 10000, threads=8)
    assert _join(bs, "seq1", "off1") == [sq.encode() for _, sq in recs] * 12
    bad = str(tmp_path / "bad.fq")
    with open(bad, "w") as f:
        f.write("@r\nACGT\nIIII\n")
    with pytest.raises(ra.QmError):
        _batches(bad, None, 10)
    with pytest.raises(ra.QmError):
        ra.FastxReader(str(tmp_path / "missing.fq"))


def test_sam_writer_reproduces_the_reference_digest(sample_data, oracle_mod):
    """native header + records on oracle hits == the md5 the reference's own SAM has on sample_data"""
    import rapmap_amd as ra
    ix, orc = load_oracle(sample_data["idx"])
    qi = ra.QuasiIndex(sample_data["idx"])
    q1, o1 = pack(sample_data["reads1"]); q2, o2 = pack(sample_data["reads2"])
    res = orc.map_pairs(q1, o1, q2, o2, nthreads=2)
    b = ra.ReadBatch(); b.n = len(o1) - 1
    b.seq1, b.off1, b.seq2, b.off2 = q1, o1, q2, o2
    nm1, no1 = pack([x.encode() for x in sample_data["names1"]]); nm2, no2 = pack([x.encode() for x in sample_data["names2"]])
    b.names1, b.name_off1, b.names2, b.name_off2 = nm1, no1, nm2, no2
    body = ra.sam_records_text(qi, b, res.hit_offsets, res.hits, threads=3)
    head = ra.sam_header_text(qi)
    text = b"".join(l for l in (head + body).splitlines(True) if not l.startswith(b"@PG"))
    want = open(os.path.join(GOLD, "sample_data", "expected_sam_body.md5")).read().strip()
    assert hashlib.md5(text).hexdigest() == want


@pytest.mark.parametrize("opts", [{}, {"maxNumHits": 3}, {"fuzzy": 1}])
def test_sam_writer_matches_python_formatter(synth_small, oracle_mod, opts):
    import rapmap_amd as ra
    import samfmt as sam
    ix, orc = load_oracle(synth_small["idx"])
    qi = ra.QuasiIndex(synth_small["idx"])
    q1, o1 = pack(synth_small["reads1"]); q2, o2 = pack(synth_small["reads2"])
    oo = oracle_mod.default_opts(**opts)
    res = orc.map_pairs(q1, o1, q2, o2, opts=oo, nthreads=4)
    b = ra.ReadBatch(); b.n = len(o1) - 1
    b.seq1, b.off1, b.seq2, b.off2 = q1, o1, q2, o2
    b.names1, b.name_off1 = pack([x.encode() for x in synth_small["names1"]])
    b.names2, b.name_off2 = pack([x.encode() for x in synth_small["names2"]])
    got = ra.sam_records_text(qi, b, res.hit_offsets, res.hits, max_num_hits=oo.maxNumHits, threads=4)
    want = "".join(sam.format_pair(synth_small["names1"][i], synth_small["reads1"][i], synth_small["names2"][i],
                                   synth_small["reads2"][i], res.hits[res.hit_offsets[i]:res.hit_offsets[i + 1]],
                                   ix.names, ix.txpLens, oo.maxNumHits) for i in range(b.n))
    assert got == want.encode()
    import tempfile
    with tempfile.TemporaryFile() as f:                       # same bytes through the file-descriptor writer
        nb = ra.sam_records_text(qi, b, res.hit_offsets, res.hits, max_num_hits=oo.maxNumHits, threads=4, fd=f.fileno())
        f.seek(0)
        assert nb == len(got) and f.read() == got
    # the run-long writer (formatting overlaps the previous batch's write): five batches behind an existing header line
    with tempfile.NamedTemporaryFile() as nf:
        nf.write(b"@HD\tpre-existing line\n"); nf.flush()
        fd = os.open(nf.name, os.O_WRONLY)
        try:
            os.lseek(fd, 0, os.SEEK_END)
            w = ra.SamWriter(qi, fd, max_num_hits=oo.maxNumHits, threads=4)
            for _ in range(5):
                w.put(b, res.hit_offsets, res.hits)
            assert w.close() == 5 * len(got)
        finally:
            os.close(fd)
        assert open(nf.name, "rb").read() == b"@HD\tpre-existing line\n" + got * 5
    # -x: gzip members compressed side by side; the file is one valid .gz stream holding header + records
    import gzip as _gz
    with tempfile.NamedTemporaryFile(suffix=".sam.gz") as nf:
        fd = os.open(nf.name, os.O_WRONLY)
        try:
            w = ra.SamWriter(qi, fd, max_num_hits=oo.maxNumHits, threads=4, gzip=True)
            w.header()
            for _ in range(3):
                w.put(b, res.hit_offsets, res.hits)
            nbz = w.close()
        finally:
            os.close(fd)
        raw = open(nf.name, "rb").read()
        assert nbz == len(raw) and raw[:2] == b"\x1f\x8b"
        assert _gz.decompress(raw) == ra.sam_header_text(qi) + got * 3
        assert len(raw) < (len(got) * 3) // 2
    # ... and a descriptor that cannot be written: the error surfaces in put or close, not as a crash
    rfd = os.open(os.devnull, os.O_RDONLY)
    try:
        w = ra.SamWriter(qi, rfd, max_num_hits=oo.maxNumHits, threads=2)
        with pytest.raises(ra.QmError):
            for _ in range(4):
                w.put(b, res.hit_offsets, res.hits)
            w.close()
        try:
            w.close()
        except ra.QmError:
            pass
    finally:
        os.close(rfd)
    # a batch large enough for several formatter threads: their parts go out in order
    if not opts:
        R = 8
        def tile(seq, off):
            return np.tile(seq[:off[-1]], R), np.concatenate([[0], (off[1:][None, :] + off[-1] * np.arange(R)[:, None]).ravel()]).astype(np.int64)
        bb = ra.ReadBatch(); bb.n = b.n * R
        bb.seq1, bb.off1 = tile(q1, o1); bb.seq2, bb.off2 = tile(q2, o2)
        bb.names1, bb.name_off1 = tile(b.names1, b.name_off1); bb.names2, bb.name_off2 = tile(b.names2, b.name_off2)
        bh, bho = tile(res.hits, res.hit_offsets)
        big = ra.sam_records_text(qi, bb, bho, bh, max_num_hits=oo.maxNumHits, threads=6)
        assert big == got * R
        with tempfile.TemporaryFile() as f:
            f.write(b"x" * 5000); f.flush()
            nb = ra.sam_records_text(qi, bb, bho, bh, max_num_hits=oo.maxNumHits, threads=6, fd=f.fileno())
            f.seek(5000)
            assert nb == len(big) and f.read() == big
    # write-only descriptor (a shell's `>`)
    with tempfile.NamedTemporaryFile() as nf:
        fd = os.open(nf.name, os.O_WRONLY)
        try:
            nb = ra.sam_records_text(qi, b, res.hit_offsets, res.hits, max_num_hits=oo.maxNumHits, threads=4, fd=fd)
        finally:
            os.close(fd)
        assert nb == len(got) and open(nf.name, "rb").read() == got
    # a descriptor opened for appending (a shell's `>>`): pwrite would ignore its offsets there, the parts must still land in order
    with tempfile.NamedTemporaryFile() as nf:
        nf.write(b"@HD\tpre-existing line\n"); nf.flush()
        fd = os.open(nf.name, os.O_WRONLY | os.O_APPEND)
        try:
            nb = ra.sam_records_text(qi, b, res.hit_offsets, res.hits, max_num_hits=oo.maxNumHits, threads=4, fd=fd)
        finally:
            os.close(fd)
        assert nb == len(got) and open(nf.name, "rb").read() == b"@HD\tpre-existing line\n" + got
    # single-end
    rs = orc.map_single(q1, o1, opts=oo, nthreads=2)
    sb = ra.ReadBatch(); sb.n = b.n; sb.seq1, sb.off1, sb.names1, sb.name_off1 = q1, o1, b.names1, b.name_off1
    got = ra.sam_records_text(qi, sb, rs.hit_offsets, rs.hits, max_num_hits=oo.maxNumHits, threads=2)
    want = "".join(sam.format_single(synth_small["names1"][i], synth_small["reads1"][i],
                                     rs.hits[rs.hit_offsets[i]:rs.hit_offsets[i + 1]], ix.names, ix.txpLens) for i in range(sb.n))
    assert got == want.encode()


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [777, 4096, 1 << 20])
def test_pipelined_stream_matches_the_oracle(sample_data, oracle_mod, batch):
    """qm_stream_*: FASTQ(.gz) -> mapped batches, reader and two device contexts running ahead of the caller; batches come back in
    input order with the oracle's hits, whatever the batch size (many small batches keep every slot of the ring in flight)"""
    import rapmap_amd as ra
    ix, orc = load_oracle(sample_data["idx"])
    qi = ra.QuasiIndex(sample_data["idx"])
    q1, o1 = pack(sample_data["reads1"]); q2, o2 = pack(sample_data["reads2"])
    res = orc.map_pairs(q1, o1, q2, o2, nthreads=2)
    st = ra.MappedStream(qi, os.path.join(GOLD, "sample_data", "reads_1.fastq.gz"), os.path.join(GOLD, "sample_data", "reads_2.fastq.gz"),
                         batch_units=batch, threads=4)
    u = 0; tot = 0; cnts = []; hits = []
    for b in st:
        assert bytes(b.seq1[: b.off1[1]]) == sample_data["reads1"][u]
        cnts.append(np.diff(b.hit_offsets).copy()); hits.append(b.hits.copy()); tot += b.counters["totHits"]; u += b.n
    st.close()
    assert u == len(o1) - 1 and tot == res.counters["totHits"]
    assert np.array_equal(np.concatenate(cnts), np.diff(res.hit_offsets))
    assert np.concatenate(hits).tobytes() == res.hits.tobytes()
    # single-end, plain text input through a temporary file
    import gzip, tempfile
    with tempfile.NamedTemporaryFile(suffix=".fq") as f:
        f.write(gzip.open(os.path.join(GOLD, "sample_data", "reads_2.fastq.gz")).read()); f.flush()
        st = ra.MappedStream(qi, f.name, None, batch_units=batch, threads=3)
        rs = orc.map_single(q2, o2, nthreads=2)
        got = np.concatenate([b.hits.copy() for b in st]) if True else None
        st.close()
        assert got.tobytes() == rs.hits.tobytes()


@pytest.mark.parametrize("chunk,batch,threads", [(777, 1, 3), (1500, 13, 1), (4096, 1000, 6), (1 << 20, 257, 4), (3000, 64, 5)])
def test_ingest_engine_chunk_boundaries(tmp_path, monkeypatch, chunk, batch, threads):
    """qm_ingest: chunks cut at arbitrary byte offsets resynchronise on record boundaries (quality lines that start with '@'
    or '+', CRLF, empty reads, names with blanks), batches span chunks, both files advance together"""
    import random
    import rapmap_amd as ra
    monkeypatch.setenv("QM_INGEST_CHUNK", str(chunk))
    if chunk == 3000:
        # copy tasks of three records: the 16-byte moves of short reads and names (their over-run must stay inside the task's own
        # stretch of the batch) next to tasks of other workers on every side
        monkeypatch.setenv("QM_INGEST_COPY_RUN", "3")
    rnd = random.Random(chunk * 31 + batch)
    n = 3000
    recs1, recs2 = [], []
    p1 = str(tmp_path / "a.fq"); p2 = str(tmp_path / "b.fq")
    with open(p1, "wb") as f1, open(p2, "wb") as f2:
        for i in range(n):
            for f, recs, nl in ((f1, recs1, b"\n"), (f2, recs2, b"\r\n" if i % 7 == 0 else b"\n")):
                L = rnd.choice([0, 1, 30, 31, 100, 100, 100, 250]) if i % 50 == 0 else 100
                if chunk == 3000:
                    L = rnd.choice([0, 1, 5, 12, 15, 16, 17, 33])         # every record shorter than a few 16-byte moves
                s = "".join(rnd.choice("ACGTN") for _ in range(L)).encode()
                q = "".join(rnd.choice("@+I5#") for _ in range(L)).encode()
                nm = ("r%d some text/%d" % (i * 7919, 1 + (f is f2))).encode() if chunk != 3000 else ("r%d" % i).encode()
                f.write(b"@" + nm + nl + s + nl + b"+" + (nm if i % 3 == 0 else b"") + nl + q + nl)
                recs.append((nm, s))
    bs = _batches(p1, p2, batch, threads=threads)
    assert all(b.n <= batch for b in bs) and sum(b.n for b in bs) == n
    assert _join(bs, "seq1", "off1") == [s for _, s in recs1] and _join(bs, "seq2", "off2") == [s for _, s in recs2]
    assert _join(bs, "names1", "name_off1") == [nm for nm, _ in recs1] and _join(bs, "names2", "name_off2") == [nm for nm, _ in recs2]
    # the same through gzip (one inflate thread per file, blocks cut on record boundaries)
    g1 = str(tmp_path / "a.fq.gz"); g2 = str(tmp_path / "b.fq.gz")
    for src, dst in ((p1, g1), (p2, g2)):
        with gzip.open(dst, "wb") as g:
            g.write(open(src, "rb").read())
    bs = _batches(g1, g2, batch, threads=threads)
    assert _join(bs, "seq1", "off1") == [s for _, s in recs1] and _join(bs, "seq2", "off2") == [s for _, s in recs2]
    assert _join(bs, "names2", "name_off2") == [nm for nm, _ in recs2]


@pytest.mark.parametrize("threads", [1, 6])
def test_ingest_engine_inflates_bgzf_blocks_in_parallel(tmp_path, monkeypatch, threads):
    """a BGZF file (independent gzip members that announce their size) is inflated by helper threads side by side and read
    exactly like the plain file and like the same bytes as one ordinary gzip stream; Python's gzip module agrees that the
    file is a valid (multi-member) gzip file"""
    import gzip
    import random
    import rapmap_amd as ra
    from util import write_bgzf
    rnd = random.Random(11)
    recs1, recs2 = [], []
    for i in range(30000):
        L = rnd.choice([1, 31, 100, 100, 100, 250, 2000]) if i % 50 == 0 else 100
        for recs, m in ((recs1, 1), (recs2, 2)):
            s = "".join(rnd.choice("ACGTN") for _ in range(L)); q = "".join(rnd.choice("@+I5#") for _ in range(L))
            recs.append(("@r%d/%d some text\n%s\n+\n%s\n" % (i, m, s, q)).encode())
    d1, d2 = b"".join(recs1), b"".join(recs2)
    plain = (str(tmp_path / "a.fq"), str(tmp_path / "b.fq")); bg = (str(tmp_path / "a.bgz.fq.gz"), str(tmp_path / "b.bgz.fq.gz"))
    gzs = (str(tmp_path / "a.fq.gz"), str(tmp_path / "b.fq.gz"))
    for data, pp, pb, pg in ((d1, plain[0], bg[0], gzs[0]), (d2, plain[1], bg[1], gzs[1])):
        open(pp, "wb").write(data)
        write_bgzf(pb, data, block=rnd.choice([777, 60000, 65280]))
        with gzip.open(pg, "wb") as f:
            f.write(data)
        assert gzip.open(pb, "rb").read() == data
    monkeypatch.setenv("QM_INGEST_BGZF_THREADS", str(threads))

    def read(p1, p2):
        out = []
        for b in _batches(p1, p2, 4096, threads=4):
            for i in range(b.n):
                out.append((bytes(b.names1[b.name_off1[i]:b.name_off1[i + 1]]), bytes(b.seq1[b.off1[i]:b.off1[i + 1]]),
                            bytes(b.names2[b.name_off2[i]:b.name_off2[i + 1]]), bytes(b.seq2[b.off2[i]:b.off2[i + 1]])))
        return out
    want = read(*plain)
    assert len(want) == 30000
    assert read(*bg) == want and read(*gzs) == want
    assert read(bg[0], gzs[1]) == want                       # one file of each kind
    monkeypatch.setenv("QM_INGEST_NO_BGZF", "1")             # the same file through the single zlib stream
    assert read(*bg) == want
    monkeypatch.delenv("QM_INGEST_NO_BGZF")
    # a block whose payload is damaged (CRC-32 mismatch), and a file cut inside a block: errors, not silence
    raw = bytearray(open(bg[0], "rb").read())
    raw[len(raw) // 2] ^= 0x55
    bad = str(tmp_path / "bad.fq.gz"); open(bad, "wb").write(bytes(raw))
    with pytest.raises(ra.QmError):
        read(bad, bg[1])
    cut = str(tmp_path / "cut.fq.gz"); open(cut, "wb").write(open(bg[0], "rb").read()[:100000])
    with pytest.raises(ra.QmError):
        read(cut, bg[1])


@pytest.mark.parametrize("kind", ["fasta", "fastq_no_final_newline", "fastq"])
def test_ingest_bgzf_slow_consumer_keeps_the_last_record(tmp_path, monkeypatch, kind):
    """a consumer slower than the inflate threads: the planner is gated by the full block queue, so the final chunk is popped
    BEFORE plan() has run into the end of the file -- its tail (for FASTA always the last record) must still come out"""
    import time
    import rapmap_amd as ra
    from util import write_bgzf
    n = 6000
    if kind == "fasta":
        data = b"".join(b">r%d\n%s\n" % (i, b"ACGT" * 25) for i in range(n))
    else:
        data = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, b"ACGT" * 25, b"I" * 100) for i in range(n))
        if kind == "fastq_no_final_newline":
            data = data[:-1]
    p1 = str(tmp_path / "a.gz"); write_bgzf(p1, data, block=3000)
    monkeypatch.setenv("QM_INGEST_BGZF_CHUNK", "20000")
    for paired in (False, True):
        rd = ra.FastxReader(p1, p1 if paired else None, threads=3)
        names = []
        for b in rd.chunks(100):
            names += [bytes(b.names1[b.name_off1[i]:b.name_off1[i + 1]]) for i in range(b.n)]
            if paired:
                assert bytes(b.seq2[: b.off2[-1]]) == bytes(b.seq1[: b.off1[-1]])
            if len(names) < 3000:
                time.sleep(0.01)
        rd.close()
        assert names == [b"r%d" % i for i in range(n)]


def test_ingest_engine_errors_and_empty_inputs(tmp_path):
    import rapmap_amd as ra
    e = str(tmp_path / "empty.fq"); open(e, "w").close()
    assert _batches(e, None, 10) == [] and _batches(e, e, 10) == []
    a = str(tmp_path / "a.fq"); b = str(tmp_path / "b.fq")
    with open(a, "w") as f:
        f.write("".join("@r%d\nACGT\n+\nIIII\n" % i for i in range(100)))
    with open(b, "w") as f:
        f.write("".join("@r%d\nACGT\n+\nIIII\n" % i for i in range(99)))
    with pytest.raises(ra.QmError, match="different numbers"):
        _batches(a, b, 10)
    with pytest.raises(ra.QmError, match="different numbers"):
        _batches(b, a, 1000)
    with pytest.raises(ra.QmError, match="different numbers"):
        _batches(a, e, 10)
    t = str(tmp_path / "trunc.fq")
    with open(t, "w") as f:
        f.write("@r0\nACGT\n+\nIIII\n@r1\nACGT\n+")
    with pytest.raises(ra.QmError, match="malformed"):
        _batches(t, None, 10)
    # a reader that is closed before it was drained (workers mid-flight) shuts down cleanly
    big = str(tmp_path / "big.fq")
    with open(big, "w") as f:
        f.write("".join("@r%d\n%s\n+\n%s\n" % (i, "ACGT" * 25, "I" * 100) for i in range(50000)))
    rd = ra.FastxReader(big, None, threads=4)
    it = rd.chunks(100)
    assert next(it).n == 100
    rd.close()
    # the batch size may shrink between calls: the current batch comes in pieces, nothing is lost or reordered
    rd = ra.FastxReader(a, None, threads=2)
    L = ra.api.lib(); import ctypes as C
    n = C.c_int64(); ptr = [C.c_void_p() for _ in range(8)]; got = []
    for want in (64, 7, 7, 100, 100, 100):
        assert L.qm_reader_next(rd._h, want, C.byref(n), *[C.byref(x) for x in ptr]) == 0
        if n.value:
            off = np.ctypeslib.as_array(C.cast(ptr[1], C.POINTER(C.c_int64)), shape=(n.value + 1,))
            nof = np.ctypeslib.as_array(C.cast(ptr[3], C.POINTER(C.c_int64)), shape=(n.value + 1,))
            assert off[0] == 0 and nof[0] == 0 and n.value <= want
            got += [C.string_at(ptr[2].value + int(nof[i]), int(nof[i + 1] - nof[i])) for i in range(n.value)]
    rd.close()
    assert got == [b"r%d" % i for i in range(100)]


@pytest.mark.gpu
@pytest.mark.parametrize("batch,names", [(611, True), (4096, False)])
def test_multi_device_stream_matches_the_oracle(synth_small, oracle_mod, tmp_path, batch, names):
    """qm_stream_open_ex over several devices (two GPUs when the box has them, otherwise the same GPU listed twice: four
    contexts, two replicas' worth of map threads): batches are dealt to the devices' contexts and still come back in input
    order with the oracle's hits and counters"""
    import torch
    import rapmap_amd as ra
    ix, orc = load_oracle(synth_small["idx"])
    qi = ra.QuasiIndex(synth_small["idx"])
    q1, o1 = pack(synth_small["reads1"]); q2, o2 = pack(synth_small["reads2"])
    res = orc.map_pairs(q1, o1, q2, o2, nthreads=4)
    f1 = str(tmp_path / "r1.fq"); f2 = str(tmp_path / "r2.fq")
    for f, nms, rds in ((f1, synth_small["names1"], synth_small["reads1"]), (f2, synth_small["names2"], synth_small["reads2"])):
        with open(f, "wb") as fh:
            for nm, r in zip(nms, rds):
                fh.write(b"@" + nm.encode() + b"\n" + r + b"\n+\n" + b"I" * len(r) + b"\n")
    devs = [0, 1] if torch.cuda.device_count() >= 2 else [0, 0]
    st = ra.MappedStream(qi, f1, f2, device=devs, batch_units=batch, threads=4, names=names)
    u = 0; cnts = []; hits = []; tot = {k: 0 for k in res.counters}; seen = set()
    for b in st:
        assert bytes(b.seq1[: b.off1[1]]) == synth_small["reads1"][u]
        if names:
            assert bytes(b.names2[: b.name_off2[1]]).decode() == synth_small["names2"][u]
        else:
            assert not hasattr(b, "names1")
        cnts.append(np.diff(b.hit_offsets).copy()); hits.append(b.hits.copy()); u += b.n; seen.add(b.device)
        for k in tot:
            tot[k] += b.counters[k]
    ss = st.stats(); st.close()
    assert ss["packed_batches"] > 0                      # the batches went to the devices 2-bit packed
    assert u == len(o1) - 1 and tot == res.counters and seen == set(devs)
    assert np.array_equal(np.concatenate(cnts), np.diff(res.hit_offsets))
    assert np.concatenate(hits).tobytes() == res.hits.tobytes()
    assert ss["bytes_parsed"] == os.path.getsize(f1) + os.path.getsize(f2)


@pytest.mark.gpu
def test_stream_batches_full_of_exceptions_travel_as_characters(synth_small, oracle_mod, tmp_path, monkeypatch):
    """a batch whose reads are mostly N / lower case outgrows the exception list of the 2-bit packing: the stream sends such a
    batch as plain characters (no error, same hits); QM_STREAM_NO_PACK=1 sends every batch that way"""
    import random
    import rapmap_amd as ra
    ix, orc = load_oracle(synth_small["idx"])
    qi = ra.QuasiIndex(synth_small["idx"])
    rnd = random.Random(2)
    r1 = list(synth_small["reads1"][:1500]); r2 = list(synth_small["reads2"][:1500])
    for i in range(0, 1500, 2):                          # every second read: a quarter of its characters lower case / N
        b = bytearray(r1[i])
        for j in range(0, len(b), 4):
            b[j] = ord("n") if rnd.random() < 0.5 else (b[j] | 0x20)
        r1[i] = bytes(b)
    q1, o1 = pack(r1); q2, o2 = pack(r2)
    res = orc.map_pairs(q1, o1, q2, o2, nthreads=4)
    f1 = str(tmp_path / "r1.fq"); f2 = str(tmp_path / "r2.fq")
    for f, rds in ((f1, r1), (f2, r2)):
        with open(f, "wb") as fh:
            for k, r in enumerate(rds):
                fh.write(b"@r%d\n" % k + r + b"\n+\n" + b"I" * len(r) + b"\n")
    for nopack in (False, True):
        if nopack:
            monkeypatch.setenv("QM_STREAM_NO_PACK", "1")
        st = ra.MappedStream(qi, f1, f2, batch_units=1500, threads=3, names=False)
        hits = np.concatenate([b.hits.copy() for b in st]); ss = st.stats(); st.close()
        assert hits.tobytes() == res.hits.tobytes()
        assert ss["packed_batches"] == 0                 # overflowed (first round) / switched off (second)


@pytest.mark.parametrize("threads,stretch", [(1, 4096), (3, 4096), (6, 20000), (4, 1 << 22)])
def test_ingest_engine_inflates_one_gzip_stream_on_several_threads(tmp_path, monkeypatch, threads, stretch):
    """an ORDINARY gzip file -- one deflate stream, what `gzip` writes and the reference reads through one zlib stream
    (src/FastxParser.cpp:229-328) -- is inflated by several threads (qm_pgz.h: guessed block starts, symbols for the unknown window,
    every guess checked against the stretch in front, CRC-32 of every member): the same records as the plain file at every
    compression level, with stored blocks, as several members in one file, with the guesses forced to fail; damaged and truncated
    files are errors"""
    import gzip
    import random
    import zlib
    import rapmap_amd as ra
    rnd = random.Random(5)
    genome = "".join(rnd.choice("ACGT") for _ in range(200000))
    recs1, recs2 = [], []
    for i in range(40000):
        L = rnd.choice([1, 31, 100, 250, 2000]) if i % 97 == 0 else 100
        for recs, m in ((recs1, 1), (recs2, 2)):
            a = rnd.randrange(0, len(genome) - L)
            sq = genome[a:a + L]; q = "".join(rnd.choice("FFFFF:,#") for _ in range(L))
            recs.append(("@SRR1.%d %d/%d\n%s\n+\n%s\n" % (i, i, m, sq, q)).encode())
    d1, d2 = b"".join(recs1), b"".join(recs2)
    plain = (str(tmp_path / "a.fq"), str(tmp_path / "b.fq"))
    open(plain[0], "wb").write(d1); open(plain[1], "wb").write(d2)
    monkeypatch.setenv("QM_INGEST_PGZ_THREADS", str(threads))
    monkeypatch.setenv("QM_PGZ_STRETCH", str(stretch))

    def read(p1, p2):
        out = []
        for b in _batches(p1, p2, 4096, threads=4):
            for i in range(b.n):
                out.append((bytes(b.names1[b.name_off1[i]:b.name_off1[i + 1]]), bytes(b.seq1[b.off1[i]:b.off1[i + 1]]),
                            bytes(b.names2[b.name_off2[i]:b.name_off2[i + 1]]), bytes(b.seq2[b.off2[i]:b.off2[i + 1]])))
        return out
    want = read(*plain)
    assert len(want) == 40000

    def gz(data, path, level, members=1):
        with open(path, "wb") as f:
            step = (len(data) + members - 1) // members
            for m in range(members):
                f.write(gzip.compress(data[m * step:(m + 1) * step], compresslevel=level))
        return path
    for level, members in ((6, 1), (1, 1), (9, 1), (0, 1), (6, 3)):
        g1 = gz(d1, str(tmp_path / ("a%d_%d.fq.gz" % (level, members))), level, members)
        g2 = gz(d2, str(tmp_path / ("b%d_%d.fq.gz" % (level, members))), level, members)
        assert read(g1, g2) == want, (level, members)
    g1 = str(tmp_path / "a6_1.fq.gz"); g2 = str(tmp_path / "b6_1.fq.gz")
    # a header with a file name and a comment, and trailing garbage behind the last member (zlib's gzread ignores it)
    raw = open(g1, "rb").read()
    fancy = str(tmp_path / "fancy.fq.gz")
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = co.compress(d1) + co.flush()
    import struct
    open(fancy, "wb").write(b"\x1f\x8b\x08\x18\x00\x00\x00\x00\x00\x03" + b"reads_1.fastq\x00" + b"a comment\x00" + body +
                            struct.pack("<II", zlib.crc32(d1) & 0xffffffff, len(d1) & 0xffffffff) + b"\x00" * 1000)
    assert read(fancy, g2) == want
    # the single zlib stream gives the same
    monkeypatch.setenv("QM_INGEST_NO_PGZ", "1")
    assert read(g1, g2) == want
    monkeypatch.delenv("QM_INGEST_NO_PGZ")
    # damaged payload (CRC-32 / symbol check), a file cut in the middle, a wrong CRC in the trailer
    bad = bytearray(raw); bad[len(bad) // 2] ^= 0x55
    p = str(tmp_path / "bad.fq.gz"); open(p, "wb").write(bytes(bad))
    with pytest.raises(ra.QmError):
        read(p, g2)
    p = str(tmp_path / "cut.fq.gz"); open(p, "wb").write(raw[:len(raw) // 2])
    with pytest.raises(ra.QmError):
        read(p, g2)
    bad = bytearray(raw); bad[-6] ^= 1
    p = str(tmp_path / "crc.fq.gz"); open(p, "wb").write(bytes(bad))
    with pytest.raises(ra.QmError):
        read(p, g2)
