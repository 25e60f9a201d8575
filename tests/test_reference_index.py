"""An index the REFERENCE wrote (tests/golden/sample_data/ref_index.tar.gz: `rapmap quasiindex` / `quasiindex -p` on
sample_data, bytes from the survey stage's probe build) through our readers: the oracle's numpy reader, qm_index_open,
and -- on the GPU -- the whole path.  Also: what qm_build_index writes for the same FASTA, file by file against it."""
import hashlib
import os
import tarfile

import numpy as np
import pytest

from conftest import GOLD, load_oracle
from util import assert_hits_equal, pack

SD = os.path.join(GOLD, "sample_data")


@pytest.fixture(scope="module")
def ref_index(tmp_path_factory):
    d = tmp_path_factory.mktemp("ref_index")
    with tarfile.open(os.path.join(SD, "ref_index.tar.gz")) as t:
        t.extractall(d)
    for line in open(os.path.join(SD, "ref_index.md5")):
        md5, name = line.split()
        assert hashlib.md5(open(d / name, "rb").read()).hexdigest() == md5, name
    return {"dense": str(d / "dense"), "perfect": str(d / "perfect")}


def test_our_indexer_writes_the_references_bytes(ref_index, sample_data, tmp_path, lib_built):
    """sa.bin, txpInfo.bin, rsd.bin and the perfect-hash files byte for byte; hash.bin as the same key -> interval map
    (sparsepp's table order depends on its growth history, the records do not)"""
    import rapmap_amd as ra
    from oracle import q5
    for fn in ("sa.bin", "txpInfo.bin", "rsd.bin"):
        assert open(os.path.join(sample_data["idx"], fn), "rb").read() == open(os.path.join(ref_index["dense"], fn), "rb").read(), fn
    ph = str(tmp_path / "ph")
    ra.build_index(os.path.join(SD, "transcripts.fasta"), ph, threads=2, perfect_hash=True)
    for fn in ("sa.bin", "txpInfo.bin", "rsd.bin", "hash_info.bph", "hash_info.val"):
        assert open(os.path.join(ph, fn), "rb").read() == open(os.path.join(ref_index["perfect"], fn), "rb").read(), fn
    a = q5.load(sample_data["idx"]); b = q5.load(ref_index["dense"])
    oa = np.argsort(a.hkeys); ob = np.argsort(b.hkeys)
    assert np.array_equal(a.hkeys[oa], b.hkeys[ob]) and np.array_equal(a.hlb[oa], b.hlb[ob]) and np.array_equal(a.hub[oa], b.hub[ob])
    assert a.hkeys.size == 18902


def test_library_opens_the_references_index(ref_index, lib_built):
    import rapmap_amd as ra
    for kind, ph in (("dense", False), ("perfect", True)):
        qi = ra.QuasiIndex(ref_index[kind])
        assert qi.k == 31 and qi.n_txps == 15 and qi.n_keys == 18902 and qi.perfect_hash == ph
        qi.close()


def test_oracle_on_the_references_index_reproduces_its_sam_digest(ref_index, sample_data, oracle_mod):
    """oracle hits on the reference-written index + SAM formatting == the md5 of the reference's own SAM"""
    import samfmt as sam
    ix, orc = load_oracle(ref_index["dense"])
    q1, o1 = pack(sample_data["reads1"]); q2, o2 = pack(sample_data["reads2"])
    res = orc.map_pairs(q1, o1, q2, o2, nthreads=2)
    body = "".join(sam.format_pair(sample_data["names1"][i], sample_data["reads1"][i], sample_data["names2"][i], sample_data["reads2"][i],
                                   res.hits[res.hit_offsets[i]:res.hit_offsets[i + 1]], ix.names, ix.txpLens) for i in range(len(o1) - 1))
    text = "".join(l for l in (sam.sam_header(ix.names, ix.txpLens) + body).splitlines(True) if not l.startswith("@PG"))
    assert hashlib.md5(text.encode()).hexdigest() == open(os.path.join(SD, "expected_sam_body.md5")).read().strip()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,compact", [("dense", False), ("perfect", False), ("perfect", True)])
def test_gpu_path_on_the_references_index(ref_index, sample_data, oracle_mod, kind, compact):
    """the bytes the reference wrote, mmap'd unchanged, through the HIP path: dense hash.bin (spp table order as the
    reference left it) and the -p files in both device images"""
    import rapmap_amd as ra
    ix, orc = load_oracle(ref_index["dense"])
    qi = ra.QuasiIndex(ref_index[kind])
    mp = ra.QuasiMapper(qi, 0, ph_compact=compact)
    q1, o1 = pack(sample_data["reads1"]); q2, o2 = pack(sample_data["reads2"])
    res = orc.map_pairs(q1, o1, q2, o2, nthreads=2)
    gr = mp.map_pairs(q1, o1, q2, o2)
    assert_hits_equal(res.hit_offsets, res.hits, gr.hit_offsets, gr.hits, "reference-written %s index" % kind)
    assert res.counters == gr.counters and abs(gr.counters["totHits"] / gr.counters["numReads"] - 1.4253) < 1e-4
