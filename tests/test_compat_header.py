"""include/qmap_compat.hpp (the C++ face with the reference's type names) compiles, links against the library and,
on the GPU box, produces the oracle's hits."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_oracle

SRC = r'''
#include <cstdio>
#include <fstream>
#include "qmap_compat.hpp"
int main(int argc, char** argv) {
  try {
    qmap::QuasiIndex ix(argv[1]);
    std::printf("k %u txps %zu ph %d\n", ix.k(), ix.txpNames.size(), (int)ix.perfectHash());
    if (argc < 4) return 0;
    qmap::QuasiMapper mp(ix, 0);
    std::vector<std::pair<std::string, std::string>> pairs;
    std::ifstream f(argv[2]); std::string a, b;
    while (f >> a >> b) pairs.emplace_back(a, b);
    std::vector<std::vector<qmap::QuasiAlignment>> joint; qmap::HitCounters hc;
    mp.mapReadPairs(pairs, joint, hc);
    std::ofstream o(argv[3]);
    for (auto& v : joint) { o << v.size(); for (auto& q : v) o << ' ' << q.tid << ':' << q.pos << ':' << q.matePos << ':' << q.fwd << q.mateIsFwd << ':' << q.fragLen << ':' << (int)q.mateStatus; o << '\n'; }
    std::printf("reads %llu tot %llu\n", (unsigned long long)hc.numReads.load(), (unsigned long long)hc.totHits.load());
  } catch (const qmap::Error& e) { std::printf("qmap error %d: %s\n", e.code(), e.what()); return 3; }
  return 0;
}
'''


def _build(tmp_path, lib_built):
    src = tmp_path / "t.cpp"; src.write_text(SRC)
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), lib_built,
                           "-Wl,-rpath," + os.path.dirname(lib_built), "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-pthread"])
    return exe


def test_compiles_links_and_opens_index(sample_data, lib_built, tmp_path):
    exe = _build(tmp_path, lib_built)
    r = subprocess.run([str(exe), sample_data["idx"]], capture_output=True, text=True)
    assert r.returncode == 0 and "k 31 txps 15 ph 0" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_cpp_caller_gets_oracle_hits(sample_data, lib_built, tmp_path, oracle_mod):
    from util import pack
    exe = _build(tmp_path, lib_built)
    n = 2000
    with open(tmp_path / "pairs.txt", "w") as f:
        for a, b in zip(sample_data["reads1"][:n], sample_data["reads2"][:n]):
            f.write(a.decode() + " " + b.decode() + "\n")
    r = subprocess.run([str(exe), sample_data["idx"], str(tmp_path / "pairs.txt"), str(tmp_path / "out.txt")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    ix, orc = load_oracle(sample_data["idx"])
    q1, o1 = pack(sample_data["reads1"][:n]); q2, o2 = pack(sample_data["reads2"][:n])
    res = orc.map_pairs(q1, o1, q2, o2)
    want = []
    for i in range(n):
        hs = res.hits[res.hit_offsets[i]:res.hit_offsets[i + 1]]
        want.append(" ".join([str(len(hs))] + ["%d:%d:%d:%d%d:%d:%d" % (h["tid"], h["pos"], h["mate_pos"], h["fwd"], h["mate_is_fwd"], h["frag_len"], h["mate_status"]) for h in hs]))
    assert open(tmp_path / "out.txt").read().splitlines() == want
    assert "reads %d tot %d" % (n, res.counters["totHits"]) in r.stdout


def test_salmon_support_members_of_quasi_alignment(tmp_path):
    """RAPMAP_SALMON_SUPPORT (the reference's include/RapMapUtils.hpp:41-43,407-421,446-467): logProb, logBias, format / libFormat()
    and fragLengthPedantic() exist and behave; the translation unit needs nothing but the header (and Salmon's LibraryFormat)."""
    exe = tmp_path / "salmon_members"
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "compat", "salmon_members.cpp"), "-o", str(exe), "-pthread"])
    assert subprocess.run([str(exe)]).returncode == 0


def test_reference_style_callers_compile_with_the_salmon_macro(tmp_path):
    """the reference-style caller and the call-surface bench compile against the header with RAPMAP_SALMON_SUPPORT defined"""
    lf = tmp_path / "LibraryFormat.hpp"
    lf.write_text("#pragma once\n#include <cstdint>\nstruct LibraryFormat { uint8_t id; static LibraryFormat formatFromID(uint8_t i) { return LibraryFormat{i}; } };\n")
    for src in ("rapmap_caller.cpp", "compat_bench.cpp"):
        subprocess.check_call(["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-DRAPMAP_SALMON_SUPPORT", "-include", str(lf), "-I", os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "tests", "compat", src)])
