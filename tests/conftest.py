import gzip
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
for _p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "emu")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def lib_built():
    """libqmap_mi355.so, built in-tree (hipcc cross-compiles without a GPU)."""
    import rapmap_amd
    # torch carries a HIP runtime of its own; it has to open the device before the library's runtime does, or its
    # initialisation fails later in the same process ("No HIP GPUs are available") -- some tests hand torch tensors to the
    # library, whatever order pytest runs them in
    if _have_gpu():
        import torch
        torch.cuda.init()
    if not os.path.exists(rapmap_amd.LIB_PATH):
        if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "rapmap_amd", "csrc")])
        else:
            pytest.fail("libqmap_mi355.so missing and no hipcc to build it")
    return rapmap_amd.LIB_PATH


def _gunzip(src, dst):
    with gzip.open(src, "rb") as g, open(dst, "wb") as o:
        shutil.copyfileobj(g, o)


@pytest.fixture(scope="session")
def sample_data(tmp_path_factory, lib_built):
    """config 1: the reference's sample_data, index built with our own quasiindex."""
    import rapmap_amd as ra
    import samfmt as sam
    d = tmp_path_factory.mktemp("sample")
    idx = str(d / "idx")
    ra.build_index(os.path.join(GOLD, "sample_data", "transcripts.fasta"), idx, threads=4)
    n1, s1 = sam.read_fastq(os.path.join(GOLD, "sample_data", "reads_1.fastq.gz"))
    n2, s2 = sam.read_fastq(os.path.join(GOLD, "sample_data", "reads_2.fastq.gz"))
    return {"idx": idx, "names1": n1, "names2": n2, "reads1": s1, "reads2": s2}


@pytest.fixture(scope="session")
def synth_small(tmp_path_factory, lib_built):
    import rapmap_amd as ra
    import samfmt as sam
    d = tmp_path_factory.mktemp("synth_small")
    fa = str(d / "txome.fa")
    _gunzip(os.path.join(GOLD, "synth_small", "txome.fa.gz"), fa)
    idx = str(d / "idx")
    ra.build_index(fa, idx, threads=4)
    n1, s1 = sam.read_fastq(os.path.join(GOLD, "synth_small", "reads_1.fastq.gz"))
    n2, s2 = sam.read_fastq(os.path.join(GOLD, "synth_small", "reads_2.fastq.gz"))
    return {"idx": idx, "fasta": fa, "names1": n1, "names2": n2, "reads1": s1, "reads2": s2}


@pytest.fixture(scope="session")
def synth_small_ph(synth_small, tmp_path_factory):
    """config 4 (small): the same transcriptome indexed with `quasiindex -p` (BooPHF / FrugalBooMap)"""
    import rapmap_amd as ra
    d = tmp_path_factory.mktemp("synth_small_ph")
    idx = str(d / "idx_ph")
    ra.build_index(synth_small["fasta"], idx, threads=4, perfect_hash=True)
    out = dict(synth_small)
    out["idx"] = idx
    return out


def _build_big(fasta, idx, **kw):
    """the indexer's int64 ("BigSA") form of an index, forced for a text that does not need it (QM_FORCE_BIGSA)"""
    import rapmap_amd as ra
    os.environ["QM_FORCE_BIGSA"] = "1"
    try:
        ra.build_index(fasta, idx, **kw)
    finally:
        del os.environ["QM_FORCE_BIGSA"]


@pytest.fixture(scope="session")
def synth_small_big(synth_small, tmp_path_factory):
    """synth_small indexed in the int64 form the reference writes for texts beyond 2^31 characters
    (src/RapMapSAIndexer.cpp:682-683,743-765): 8-byte suffix array entries, transcript starts and interval bounds"""
    d = tmp_path_factory.mktemp("synth_small_big")
    out = dict(synth_small)
    out["idx"] = str(d / "idx_big")
    _build_big(synth_small["fasta"], out["idx"], threads=4)
    return out


@pytest.fixture(scope="session")
def synth_small_big_ph(synth_small, tmp_path_factory):
    d = tmp_path_factory.mktemp("synth_small_big_ph")
    out = dict(synth_small)
    out["idx"] = str(d / "idx_big_ph")
    _build_big(synth_small["fasta"], out["idx"], threads=4, perfect_hash=True)
    return out


@pytest.fixture(scope="session")
def synth_medium(tmp_path_factory, lib_built):
    """~1/40 of config 2: 1000 genes (~5k transcripts, ~8 M chars), 60k pairs 2x100 bp, 1 % errors."""
    import rapmap_amd as ra
    from rapmap_amd import synth
    d = tmp_path_factory.mktemp("synth_medium")
    names, txps = synth.make_transcriptome(1000, seed=42, paralog_frac=0.05)
    fa = str(d / "txome.fa")
    synth.write_fasta(fa, names, txps)
    idx = str(d / "idx")
    ra.build_index(fa, idx, threads=8)
    s1, s2, off, truth = synth.make_reads(txps, 60000, seed=43)
    return {"idx": idx, "seq1": s1, "seq2": s2, "off": off, "txps": txps, "fasta": fa}


@pytest.fixture(scope="session")
def synth_medium_ph(synth_medium, tmp_path_factory):
    """the medium transcriptome indexed with `quasiindex -p`"""
    import rapmap_amd as ra
    d = tmp_path_factory.mktemp("synth_medium_ph")
    idx = str(d / "idx_ph")
    ra.build_index(synth_medium["fasta"], idx, threads=8, perfect_hash=True)
    out = dict(synth_medium)
    out["idx"] = idx
    return out


@pytest.fixture(scope="session")
def repeat_data(tmp_path_factory, lib_built):
    """repeat families of 40 / 300 / 900 / 1100 copies of a 200-mer core with unique flanks, reads drawn inside the cores
    (900 copies: below maxInterval, so with -s a strand's five or more capped MMPs bring > 4096 suffixes: the slow pass)"""
    import rapmap_amd as ra
    from rapmap_amd import synth
    rng = np.random.default_rng(99)
    B = np.frombuffer(b"ACGT", dtype=np.uint8)
    comp = np.zeros(256, np.uint8); comp[list(b"ACGT")] = list(b"TGCA")
    names, txps = synth.make_transcriptome(40, seed=5)
    cores = []
    for fam, copies in (("F40", 40), ("F300", 300), ("F1100", 1100), ("F900", 900)):
        core = B[rng.integers(0, 4, 200)]
        cores.append(core)
        for i in range(copies):
            l = B[rng.integers(0, 4, int(rng.integers(30, 80)))]; r = B[rng.integers(0, 4, int(rng.integers(30, 80)))]
            txps.append(np.concatenate([l, core, r])); names.append("%s.%d" % (fam, i))
    d = tmp_path_factory.mktemp("repeats")
    fa = str(d / "t.fa"); synth.write_fasta(fa, names, txps)
    idx = str(d / "idx"); ra.build_index(fa, idx, threads=4)
    r1, r2 = [], []
    for core in cores:
        for j in range(12):
            a = core[j:j + 100].copy(); b = comp[core[200 - 100 - j:200 - j][::-1]]
            if j % 3 == 0:
                a[50] = B[(np.searchsorted(B, a[50]) + 1) % 4]      # one substitution: two MMPs
            if j % 2:
                a, b = b, a
            r1.append(a.tobytes()); r2.append(b.tobytes())
    # a few ordinary pairs and one read crossing from a unique flank into a core
    s1, s2, off, _ = synth.make_reads(txps[:150], 200, seed=8)
    r1 += [s1[off[i]:off[i + 1]].tobytes() for i in range(200)]
    r2 += [s2[off[i]:off[i + 1]].tobytes() for i in range(200)]
    t = txps[-1]
    r1.append(t[10:110].tobytes()); r2.append(comp[t[120:220][::-1]].tobytes())
    return {"idx": idx, "reads1": r1, "reads2": r2}


@pytest.fixture(scope="session")
def runs_data(tmp_path_factory, lib_built):
    """transcripts with homopolymer runs of 24 .. 40 bases (and dinucleotide repeats) inside them; reads start 0 .. 70 bases
    before a run, so the run sits at every alignment of the four-characters-per-lane setup and windows of exactly k equal
    bases exist or just do not (isHomoPolymer, SACollector.hpp:498-536 / Kmer.hpp:484-487)"""
    import rapmap_amd as ra
    from rapmap_amd import synth
    rng = np.random.default_rng(4242)
    B = np.frombuffer(b"ACGT", dtype=np.uint8)
    comp = np.zeros(256, np.uint8); comp[list(b"ACGT")] = list(b"TGCA")
    names, txps, spots = [], [], []
    for i, run in enumerate(list(range(24, 41)) + [31, 31, 32, 30, 64, 90]):
        base = B[i % 4]
        mid = np.full(run, base, np.uint8)
        if i % 7 == 3:                                   # dinucleotide repeat instead of a homopolymer
            mid = np.tile(np.array([base, B[(i + 1) % 4]], np.uint8), (run + 1) // 2)[:run]
        l = B[rng.integers(0, 4, 150)]; r = B[rng.integers(0, 4, 150)]
        if l[-1] == base: l[-1] = B[(i + 1) % 4]
        if r[0] == base: r[0] = B[(i + 2) % 4]
        txps.append(np.concatenate([l, mid, r])); names.append("run%d_%d" % (run, i)); spots.append((150, run))
    d = tmp_path_factory.mktemp("runs")
    fa = str(d / "t.fa"); synth.write_fasta(fa, names, txps)
    idx = str(d / "idx"); ra.build_index(fa, idx, threads=2)
    r1, r2 = [], []
    for t, (st, run) in zip(txps, spots):
        for back in list(range(0, 8)) + [17, 30, 45, 69, 70, 71, 99]:
            a0 = max(0, st - back)
            a = t[a0:a0 + 100].copy()
            b0 = min(len(t) - 100, a0 + 120)
            b = comp[t[b0:b0 + 100][::-1]]
            if back % 5 == 4:
                a[min(99, back + run + 3)] = B[(np.searchsorted(B, a[min(99, back + run + 3)]) + 1) % 4]
            if back % 2:
                a, b = b, a
            r1.append(a.tobytes()); r2.append(b.tobytes())
    r1.append(b"A" * 100); r2.append(b"T" * 100)          # nothing but one base
    r1.append(b"A" * 31 + t[:69].tobytes()); r2.append(b"C" * 30 + t[:70].tobytes())
    return {"idx": idx, "reads1": r1, "reads2": r2}


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle


def load_oracle(idx_dir):
    from oracle import oracle, q5
    ix = q5.load(idx_dir)
    return ix, oracle.Oracle(ix)
