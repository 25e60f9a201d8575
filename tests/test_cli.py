"""`python -m rapmap_amd quasiindex|quasimap`: the reference's CLI surface (config 1 is exactly this: sample_data
through the command line)."""
import hashlib
import os
import subprocess
import sys

import pytest

from conftest import GOLD, ROOT

SD = os.path.join(GOLD, "sample_data")


def _run(args, **kw):
    return subprocess.run([sys.executable, "-m", "rapmap_amd"] + args, cwd=ROOT, capture_output=True, text=True, **kw)


def test_quasiindex_cli_matches_api(sample_data, tmp_path):
    r = _run(["quasiindex", "-t", os.path.join(SD, "transcripts.fasta"), "-i", str(tmp_path / "idx"), "-k", "31"])
    assert r.returncode == 0, r.stderr
    for fn in ("sa.bin", "txpInfo.bin", "rsd.bin", "hash.bin", "header.json"):
        assert open(tmp_path / "idx" / fn, "rb").read() == open(os.path.join(sample_data["idx"], fn), "rb").read(), fn
    r = _run(["quasiindex", "-t", os.path.join(SD, "transcripts.fasta"), "-i", str(tmp_path / "idx2"), "-k", "30"])
    assert r.returncode != 0 and "odd" in r.stderr


def test_quasimap_cli_validation(sample_data):
    r = _run(["quasimap", "-i", sample_data["idx"]])
    assert r.returncode != 0 and "paired-end" in r.stderr
    r = _run(["quasimap", "-i", sample_data["idx"], "-1", "a", "-2", "b", "-r", "c"])
    assert r.returncode != 0 and "not both" in r.stderr
    r = _run(["quasimap", "-i", sample_data["idx"], "-r", "x.fq", "--recoverOrphans"])
    assert r.returncode != 0 and "recoverOrphans" in r.stderr
    r = _run(["frobnicate"])
    assert r.returncode != 0 and "not yet implemented" in r.stderr


@pytest.mark.gpu
def test_quasimap_cli_sample_data_sam(sample_data, tmp_path):
    """config 1: `quasimap -i idx -1 reads_1.fastq -2 reads_2.fastq -o out.sam` reproduces the reference's SAM"""
    out = tmp_path / "out.sam"
    r = _run(["quasimap", "-i", sample_data["idx"], "-1", os.path.join(SD, "reads_1.fastq.gz"), "-2",
              os.path.join(SD, "reads_2.fastq.gz"), "-o", str(out), "-t", "2"])
    assert r.returncode == 0, r.stderr
    assert "Final # hits per read = 1.4253" in r.stderr
    text = "".join(l for l in open(out) if not l.startswith("@PG"))
    want = open(os.path.join(SD, "expected_sam_body.md5")).read().strip()
    assert hashlib.md5(text.encode()).hexdigest() == want
    # -x: the same text as a gzip stream (members compressed in parallel by the writer), to a file and to stdout
    import gzip
    outz = tmp_path / "out.sam.gz"
    r = _run(["quasimap", "-i", sample_data["idx"], "-1", os.path.join(SD, "reads_1.fastq.gz"), "-2",
              os.path.join(SD, "reads_2.fastq.gz"), "-o", str(outz), "-t", "3", "-x"])
    assert r.returncode == 0, r.stderr
    assert gzip.open(outz, "rb").read() == open(out, "rb").read()


@pytest.mark.gpu
def test_quasimap_cli_single_end_and_flags(sample_data, tmp_path, oracle_mod):
    from conftest import load_oracle
    import samfmt as sam
    from util import pack
    out = tmp_path / "se.sam"
    r = _run(["quasimap", "-i", sample_data["idx"], "-r", os.path.join(SD, "reads_1.fastq.gz"), "-o", str(out), "-m", "2", "-q"])
    assert r.returncode == 0, r.stderr
    ix, orc = load_oracle(sample_data["idx"])
    q, o = pack(sample_data["reads1"])
    res = orc.map_single(q, o, opts=oracle_mod.default_opts(maxNumHits=2), nthreads=2)
    want = sam.sam_header(ix.names, ix.txpLens) + "".join(
        sam.format_single(sample_data["names1"][i], sample_data["reads1"][i], res.hits[res.hit_offsets[i]:res.hit_offsets[i + 1]],
                          ix.names, ix.txpLens) for i in range(len(o) - 1))
    assert open(out).read() == want
    r = _run(["quasimap", "-i", sample_data["idx"], "-r", os.path.join(SD, "reads_1.fastq.gz"), "-n", "--recoverOrphans"])
    assert r.returncode != 0 and "not implemented" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("variant,flags", [("selAln", ["-s"]), ("selAln_hardFilter", ["-s", "--hardFilter"]), ("mimicBT2", ["--mimicBT2"])])
def test_quasimap_cli_selective_alignment(sample_data, tmp_path, variant, flags):
    """config 5 through the CLI == the reference's SAM (records incl. AS:i, SEQ column dropped in the fixture)"""
    import gzip
    out = tmp_path / "s.sam"
    r = _run(["quasimap", "-i", sample_data["idx"], "-1", os.path.join(SD, "reads_1.fastq.gz"), "-2",
              os.path.join(SD, "reads_2.fastq.gz"), "-o", str(out), "-q"] + flags)
    assert r.returncode == 0, r.stderr
    got = []
    for l in open(out):
        if l[0] == "@":
            continue
        c = l.split("\t")
        got.append("\t".join(c[:9] + c[10:]))
    want = gzip.open(os.path.join(SD, "expected_%s.noseq.sam.gz" % variant), "rt").read()
    assert "".join(got) == want
