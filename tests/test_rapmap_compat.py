"""Face 2 of the boundary: include/qmap_rapmap_compat.hpp gives RapMap's own call surface -- SACollector::operator(),
hit_manager::hitsToMappingsSimple, utils::mergeLeftRightHits[Fuzzy] with the reference's argument lists -- filled from the
GPU library.  tests/compat/rapmap_caller.cpp is written against the reference's call sequence
(src/RapMapSAMapper.cpp:466-551) and compiles against that header alone; on the GPU box it must produce the oracle's
jointHits, from a prefetched chunk and from batches of one alike."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_oracle
from util import pack

SRC = os.path.join(ROOT, "tests", "compat", "rapmap_caller.cpp")


@pytest.fixture(scope="module")
def caller(tmp_path_factory, lib_built):
    d = tmp_path_factory.mktemp("compat")
    exe = str(d / "rapmap_caller")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe, lib_built,
                           "-Wl,-rpath," + os.path.dirname(lib_built), "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-pthread"])
    return exe


def test_reference_style_caller_compiles_against_the_header_alone(caller, sample_data):
    """no include of anything but qmap_rapmap_compat.hpp; opening an index needs no GPU"""
    text = open(SRC).read()
    assert [l for l in text.splitlines() if l.startswith('#include "')] == ['#include "qmap_rapmap_compat.hpp"']
    r = subprocess.run([caller, sample_data["idx"]], capture_output=True, text=True)
    assert r.returncode == 0 and "k 31 txps 15 ph 0" in r.stdout, r.stdout + r.stderr


def test_caller_opens_a_bigsa_index_with_the_int64_instantiation(caller, synth_small_big):
    """RapMapSAIndex<int64_t, RegHashT> (SAIndex64BitDense, src/HitManager.cpp:889-892), picked from header.json"""
    r = subprocess.run([caller, synth_small_big["idx"]], capture_output=True, text=True)
    assert r.returncode == 0 and "ph 0 big 1" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["plain", "chain"])
def test_int64_instantiation_reproduces_the_oracles_joint_hits(caller, synth_small, synth_small_big, oracle_mod, tmp_path, mode):
    """the reference-style caller on SACollector<RapMapSAIndex<int64_t, ...>> / SAIntervalHit<int64_t> == the int64 oracle"""
    sd = synth_small_big
    keep = [i for i in range(len(sd["reads1"])) if b" " not in sd["reads1"][i] and b" " not in sd["reads2"][i]][:3000]
    r1 = [sd["reads1"][i] for i in keep]; r2 = [sd["reads2"][i] for i in keep]
    _write_pairs(tmp_path / "pairs.txt", r1, r2)
    if mode == "plain":
        ix, orc = load_oracle(sd["idx"])
        res = orc.map_pairs(*pack(r1), *pack(r2), nthreads=4)
        assert _run(caller, sd["idx"], tmp_path / "pairs.txt", tmp_path / "out.txt") == _want(res, len(r1))
        assert _run(caller, sd["idx"], tmp_path / "pairs.txt", tmp_path / "out2.txt", "--no-prefetch", "--edit") == \
            _run(caller, synth_small["idx"], tmp_path / "pairs.txt", tmp_path / "out3.txt", "--no-prefetch", "--edit")
    else:
        assert _run(caller, sd["idx"], tmp_path / "pairs.txt", tmp_path / "a.txt", "--chain") == \
            _run(caller, synth_small["idx"], tmp_path / "pairs.txt", tmp_path / "b.txt", "--chain")


def _write_pairs(path, r1, r2):
    with open(path, "w") as f:
        for a, b in zip(r1, r2):
            f.write((a.decode() or "-") + " " + (b.decode() or "-") + "\n")


def _want(res, n):
    out = []
    for i in range(n):
        hs = res.hits[res.hit_offsets[i]:res.hit_offsets[i + 1]]
        out.append(" ".join([str(len(hs))] + ["%d:%d:%d:%d%d:%d:%d" % (h["tid"], h["pos"], h["mate_pos"], h["fwd"], h["mate_is_fwd"], h["frag_len"], h["mate_status"]) for h in hs]))
    c = res.counters
    out.append("counters %d %d %d %d %d" % (c["peHits"], c["seHits"], c["totHits"], c["numReads"], c["tooManyHits"]))
    return out


def _run(caller, idx, pairs, out, *flags):
    r = subprocess.run([caller, idx, str(pairs), str(out)] + list(flags), capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout + r.stderr
    return open(out).read().splitlines()


def _clean(reads):
    # the pairs file is whitespace separated: keep reads without blanks (all of them here) and map empty reads to "-"
    return [r for r in reads]


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["sample_data", "synth_small"])
@pytest.mark.parametrize("mode", ["plain", "fuzzy", "noOrphans_m3", "fuzzy_m3"])
def test_caller_reproduces_the_oracles_joint_hits(caller, sample_data, synth_small, oracle_mod, tmp_path, which, mode):
    sd = sample_data if which == "sample_data" else synth_small
    keep = [i for i in range(len(sd["reads1"])) if b" " not in sd["reads1"][i] and b" " not in sd["reads2"][i]][:6000]
    r1 = [sd["reads1"][i] for i in keep]; r2 = [sd["reads2"][i] for i in keep]
    _write_pairs(tmp_path / "pairs.txt", r1, r2)
    oo, flags = {"plain": ({}, []), "fuzzy": ({"fuzzy": 1}, ["--fuzzy"]), "noOrphans_m3": ({"noOrphans": 1, "maxNumHits": 3}, ["--noOrphans", "--maxNumHits", "3"]),
                 "fuzzy_m3": ({"fuzzy": 1, "maxNumHits": 3}, ["--fuzzy", "--maxNumHits", "3"])}[mode]
    ix, orc = load_oracle(sd["idx"])
    q1, o1 = pack(r1); q2, o2 = pack(r2)
    res = orc.map_pairs(q1, o1, q2, o2, opts=oracle_mod.default_opts(**oo), nthreads=4)
    want = _want(res, len(r1))
    got = _run(caller, sd["idx"], tmp_path / "pairs.txt", tmp_path / "out.txt", *flags)
    assert got == want, "prefetched chunk: first difference at line %d" % next(i for i, (a, b) in enumerate(zip(got, want)) if a != b)
    # round 6: a collector that said it never looks inside HitCollectorInfo (setKeepIntervals(false): no interval records come down, the
    # pass runs on the pair / lean kernels) hands out the same joint hits; so do reads that cross PCIe as characters (QMAP_COMPAT_NO_PACK)
    got_ni = _run(caller, sd["idx"], tmp_path / "pairs.txt", tmp_path / "out_ni.txt", "--no-intervals", *flags)
    assert got_ni == want, "without interval records: first difference at line %d" % next(i for i, (a, b) in enumerate(zip(got_ni, want)) if a != b)
    if mode == "plain":
        os.environ["QMAP_COMPAT_NO_PACK"] = "1"
        try:
            got_np = _run(caller, sd["idx"], tmp_path / "pairs.txt", tmp_path / "out_np.txt", *flags)
        finally:
            del os.environ["QMAP_COMPAT_NO_PACK"]
        assert got_np == want
    # the same calls as batches of one (no prefetch): a few hundred pairs, one GPU launch per call
    m = 300
    _write_pairs(tmp_path / "few.txt", r1[:m], r2[:m])
    resf = orc.map_pairs(*pack(r1[:m]), *pack(r2[:m]), opts=oracle_mod.default_opts(**oo), nthreads=2)
    gotf = _run(caller, sd["idx"], tmp_path / "few.txt", tmp_path / "outf.txt", "--no-prefetch", *flags)
    assert gotf == _want(resf, m)


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [[], ["--fuzzy"], ["--chain"]])
def test_edited_intervals_and_chaining_agree_between_chunk_and_single_calls(caller, synth_small, tmp_path, flags):
    """a caller that edits hcInfo between the collector and hitsToMappingsSimple gets the edited intervals mapped (the chunk's
    answer is not used for that read); the chaining configuration (doChaining + considerMultiPos) gives the same jointHits
    from a chunk and from single calls"""
    r1 = synth_small["reads1"][:400]; r2 = synth_small["reads2"][:400]
    keep = [i for i in range(len(r1)) if b" " not in r1[i] and b" " not in r2[i]]
    _write_pairs(tmp_path / "p.txt", [r1[i] for i in keep], [r2[i] for i in keep])
    for extra in ([], ["--edit"]):
        a = _run(caller, synth_small["idx"], tmp_path / "p.txt", tmp_path / "a.txt", *flags, *extra)
        b = _run(caller, synth_small["idx"], tmp_path / "p.txt", tmp_path / "b.txt", "--no-prefetch", *flags, *extra)
        assert a == b, (flags, extra)
    plain = _run(caller, synth_small["idx"], tmp_path / "p.txt", tmp_path / "c.txt", *flags)
    edited = _run(caller, synth_small["idx"], tmp_path / "p.txt", tmp_path / "d.txt", *flags, "--edit")
    assert plain != edited, "dropping intervals changed nothing: the edit was not looked at"


@pytest.mark.gpu
def test_refilled_buffers_and_out_of_order_calls_never_get_a_stale_chunk_entry(caller, sample_data, oracle_mod, tmp_path):
    """the chunk recognises its reads by address, length AND content, and looks a little ahead when the caller skipped reads:
    a string buffer that was refilled in place after the prefetch is mapped on its own (not answered with what used to be
    there), and a caller that walks the chunk 0, 2, 4, ... and then 1, 3, 5, ... gets every read's own answer"""
    sd = sample_data
    keep = [i for i in range(len(sd["reads1"])) if b" " not in sd["reads1"][i] and b" " not in sd["reads2"][i]][:3000]
    r1 = [sd["reads1"][i] for i in keep]; r2 = [sd["reads2"][i] for i in keep]
    _write_pairs(tmp_path / "pairs.txt", r1, r2)
    ix, orc = load_oracle(sd["idx"])
    # what the caller's --refill does to the left reads (chunks of 5000 pairs: one chunk here)
    m1 = list(r1)
    changed = 0
    for i in range(0, len(m1) - 1, 7):
        if len(m1[i]) == len(r1[i + 1]):
            changed += m1[i] != r1[i + 1]
            m1[i] = r1[i + 1]
    assert changed > 100
    res = orc.map_pairs(*pack(m1), *pack(r2), nthreads=4)
    got = _run(caller, sd["idx"], tmp_path / "pairs.txt", tmp_path / "out.txt", "--refill")
    assert got == _want(res, len(r1))
    res0 = orc.map_pairs(*pack(r1), *pack(r2), nthreads=4)
    got = _run(caller, sd["idx"], tmp_path / "pairs.txt", tmp_path / "out2.txt", "--evens-first")
    assert got == _want(res0, len(r1))


@pytest.mark.gpu
@pytest.mark.parametrize("threads,chunk", [(1, 10000), (4, 1000), (16, 700)])
def test_reference_call_surface_under_worker_threads(synth_medium, oracle_mod, tmp_path, threads, chunk):
    """tests/compat/compat_bench.cpp -- T worker threads taking read groups from a shared hand-out and running the reference's
    per-pair sequence through the header (what bench.py's `compat_face` leg times): every jointHits vector and the HitCounters
    equal the oracle's, whatever the thread count and group size (contexts per thread, chunks in flight side by side)"""
    import json
    import bench
    sd = synth_medium
    ix, orc = load_oracle(sd["idx"])
    res = orc.map_pairs(sd["seq1"], sd["off"], sd["seq2"], sd["off"], nthreads=4)
    n = len(sd["off"]) - 1; L = int(sd["off"][1])
    assert np.array_equal(np.diff(sd["off"]), np.full(n, L))
    exe = bench.build_compat_bench(str(tmp_path))
    rp = str(tmp_path / "reads.bin")
    with open(rp, "wb") as f:
        f.write(np.asarray(sd["seq1"][: n * L]).tobytes()); f.write(np.asarray(sd["seq2"][: n * L]).tobytes())
    r = subprocess.run([exe, sd["idx"], rp, str(n), str(L), str(threads), str(chunk)], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout + r.stderr
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    c = res.counters
    assert j["digest"] == bench.compat_digest(res.hit_offsets, res.hits)
    assert (j["peHits"], j["seHits"], j["totHits"], j["numReads"], j["tooManyHits"]) == (c["peHits"], c["seHits"], c["totHits"], c["numReads"], c["tooManyHits"])
    # a shorter prefix of the same file (--use): units keep their numbers
    r = subprocess.run([exe, sd["idx"], rp, str(n), str(L), "2", "512", "--use", "5000"], capture_output=True, text=True, timeout=1200)
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["pairs"] == 5000 and j["digest"] == bench.compat_digest(res.hit_offsets[:5001], res.hits)
    # groups a worker has on their way: prefetch() alone (--depth 1), and three sent ahead with prefetch_async (the default is two)
    for depth in ("1", "3"):
        r = subprocess.run([exe, sd["idx"], rp, str(n), str(L), "3", "300", "--use", "7000", "--depth", depth], capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, r.stdout + r.stderr
        j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert j["pairs"] == 7000 and j["digest"] == bench.compat_digest(res.hit_offsets[:7001], res.hits), depth


def test_compat_digest_is_order_and_field_sensitive():
    import bench
    from rapmap_amd.api import HIT_DTYPE
    h = np.zeros(3, dtype=HIT_DTYPE); h["tid"] = [1, 2, 3]; h["pos"] = [5, -6, 7]; h["mate_status"] = [3, 1, 2]; h["fwd"] = 1; h["mate_is_fwd"] = 1
    off = np.array([0, 2, 3]); d0 = bench.compat_digest(off, h)
    assert d0 != bench.compat_digest(np.array([0, 1, 3]), h) and d0 != bench.compat_digest(off, h[[1, 0, 2]])
    g = h.copy(); g["mate_pos"][1] = 9                    # an orphan's matePos is not part of the record
    assert d0 == bench.compat_digest(off, g)
    g = h.copy(); g["mate_pos"][0] = 9
    assert d0 != bench.compat_digest(off, g)


UNIT = r"""
#include <cstdio>
#include <thread>
#include <vector>
#include <algorithm>
#include "qmap_rapmap_compat.hpp"
#define CHECK(x) do { if (!(x)) { std::printf("FAILED line %d: %s\n", __LINE__, #x); return 1; } } while (0)
int main() {
  // qmap::small_vector: inline up to 16, heap beyond, copies / moves, what the reference's code does with its position lists
  qmap::small_vector<int32_t> a;
  CHECK(a.empty() && a.capacity() == 16);
  for (int i = 0; i < 16; ++i) a.push_back(100 - i);
  const int32_t* inl = a.data();
  CHECK(a.size() == 16 && a.front() == 100 && a.back() == 85);
  a.push_back(7); CHECK(a.size() == 17 && a.data() != inl && a[16] == 7 && a[3] == 97);
  std::sort(a.begin(), a.end()); CHECK(a.front() == 7 && a.back() == 100);
  qmap::small_vector<int32_t> b = a; CHECK(b == a && b.data() != a.data());
  qmap::small_vector<int32_t> c = std::move(a); CHECK(c == b && a.empty() && a.capacity() == 16);
  a.push_back(1); CHECK(a.size() == 1 && c.size() == 17);
  qmap::small_vector<int32_t> d{1, 2, 3}; d.insert(d.begin() + 1, 9); d.erase(d.begin()); CHECK(d.size() == 3 && d[0] == 9 && d[1] == 2 && d[2] == 3);
  qmap::small_vector<int32_t> e = std::move(d); CHECK(e.size() == 3 && e[0] == 9 && d.empty());      // inline contents move by copy
  b = e; CHECK(b.size() == 3 && b[2] == 3); b.resize(5, -1); CHECK(b[4] == -1); b.clear(); CHECK(b.empty());
  rapmap::utils::QuasiAlignment q(5, 17, true, 100); q.allPositions.push_back(17);
  std::vector<rapmap::utils::QuasiAlignment> v; for (int i = 0; i < 100; ++i) v.push_back(q);        // reallocation moves the elements
  CHECK(v[99].allPositions.size() == 1 && v[0].allPositions[0] == 17 && v[50].tid == 5);
  // HitCounters: the reference's member names, counted per thread, summed when read
  rapmap::utils::HitCounters hc;
  std::vector<std::thread> th;
  for (int t = 0; t < 8; ++t) th.emplace_back([&hc] { for (int i = 0; i < 100000; ++i) { ++hc.numReads; hc.totHits += 3; hc.peHits++; } });
  for (auto& t : th) t.join();
  CHECK(hc.numReads.load() == 800000 && hc.totHits == 2400000u && hc.peHits.load() == 800000 && hc.seHits.load() == 0);
  hc.lastPrint.store(42); CHECK(hc.lastPrint.load() == 42); hc.lastPrint = 7; CHECK(uint64_t(hc.lastPrint) == 7);
  CHECK(hc.tooManyHits.fetch_add(5) == 0 && hc.tooManyHits.load() == 5);
  std::printf("ok\n");
  return 0;
}
"""


def test_header_containers_behave(tmp_path, lib_built):
    """the host-side pieces of include/qmap_rapmap_compat.hpp that replaced reference types for speed -- qmap::small_vector (for
    chobo::small_vector) and the sharded HitCounters (for std::atomic members) -- do what the reference's code expects of them"""
    src = tmp_path / "unit.cpp"; src.write_text(UNIT)
    exe = tmp_path / "unit"
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), lib_built,
                           "-Wl,-rpath," + os.path.dirname(lib_built), "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-pthread"])
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr


@pytest.mark.gpu
def test_worker_threads_with_different_settings_share_the_service(synth_medium, oracle_mod, tmp_path):
    """compat_bench --mixed: even read groups are mapped plain, odd ones with the fuzzy merge -- groups of different settings
    arrive at the batching service interleaved from 6 threads and must never ride in one batch: every jointHits vector equals
    the oracle's under the settings of its group"""
    import json
    import bench
    sd = synth_medium
    ix, orc = load_oracle(sd["idx"])
    n = len(sd["off"]) - 1; L = int(sd["off"][1]); chunk = 500
    plain = orc.map_pairs(sd["seq1"], sd["off"], sd["seq2"], sd["off"], nthreads=4)
    fuzzy = orc.map_pairs(sd["seq1"], sd["off"], sd["seq2"], sd["off"], opts=oracle_mod.default_opts(fuzzy=1), nthreads=4)
    # the expected hits: unit u takes the fuzzy result when its group (u // chunk) is odd
    use_f = ((np.arange(n) // chunk) & 1) == 1
    cnt = np.where(use_f, np.diff(fuzzy.hit_offsets), np.diff(plain.hit_offsets))
    off = np.zeros(n + 1, dtype=np.int64); off[1:] = np.cumsum(cnt)
    hits = np.zeros(int(off[-1]), dtype=plain.hits.dtype)
    for u in range(n):
        src = fuzzy if use_f[u] else plain
        hits[off[u]:off[u + 1]] = src.hits[src.hit_offsets[u]:src.hit_offsets[u + 1]]
    exe = bench.build_compat_bench(str(tmp_path))
    rp = str(tmp_path / "reads.bin")
    with open(rp, "wb") as f:
        f.write(np.asarray(sd["seq1"][: n * L]).tobytes()); f.write(np.asarray(sd["seq2"][: n * L]).tobytes())
    r = subprocess.run([exe, sd["idx"], rp, str(n), str(L), "6", str(chunk), "--mixed"], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout + r.stderr
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["digest"] == bench.compat_digest(off, hits)
