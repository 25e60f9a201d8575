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
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle


def load_oracle(idx_dir):
    from oracle import oracle, q5
    ix = q5.load(idx_dir)
    return ix, oracle.Oracle(ix)
