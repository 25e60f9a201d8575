"""The C-ABI library loads without a GPU, exports every symbol include/qmap_mi355.h declares,
and fails loudly (no CPU fallback) when there is no device."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT, _have_gpu


def test_symbols_exported(lib_built):
    import rapmap_amd as ra
    L = C.CDLL(lib_built)
    hdr = open(os.path.join(ROOT, "include", "qmap_mi355.h")).read()
    declared = set(re.findall(r"\b(qm_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(ra.ABI_SYMBOLS)
    for s in declared:
        assert hasattr(L, s), s


def test_defaults_match_reference_cli(lib_built):
    import rapmap_amd as ra
    o = ra.default_opts()
    # src/RapMapSAMapper.cpp:992-1023,1113-1114 ; include/SACollector.hpp:77
    assert (o.sensitive, o.strict_check, o.max_num_hits, o.no_orphans, o.no_dovetail, o.fuzzy, o.max_interval,
            o.sel_aln, o.quasi_cov) == (1, 1, 200, 0, 0, 0, 1000, 0, 0.0)
    # -s sub-options, src/RapMapSAMapper.cpp:1012-1023
    assert (o.hard_filter, o.match_score, o.mismatch_penalty, o.gap_open, o.gap_extend, o.dp_bandwidth, o.max_mmp_extension,
            o.aln_policy, o.min_score_fraction, o.consensus_slack) == (0, 2, -4, 4, 2, 15, 7, 0, 0.65, 0.2)


def test_struct_sizes(lib_built):
    import rapmap_amd as ra
    assert ra.HIT_DTYPE.itemsize == 32 and ra.INTERVAL_DTYPE.itemsize == 20
    assert C.sizeof(ra.QmOpts) == 88      # 8 x i32 + f64 + 8 x i32 + 2 x f64


def test_index_open_errors(lib_built, tmp_path):
    import rapmap_amd as ra
    with pytest.raises(ra.QmError, match="header.json"):
        ra.QuasiIndex(str(tmp_path / "nope"))
    d = tmp_path / "bad"
    d.mkdir()
    (d / "header.json").write_text('{"value0": {"IndexVersion": "q5", "KmerLen": 31, "BigSA": true, "PerfectHash": false}}')
    with pytest.raises(ra.QmError, match="sa.bin"):          # a BigSA header is accepted (tests/test_bigsa.py); the files are missing
        ra.QuasiIndex(str(d))
    (d / "header.json").write_text('{"value0": {"IndexVersion": "q4", "KmerLen": 31, "BigSA": false, "PerfectHash": false}}')
    with pytest.raises(ra.QmError, match="version"):
        ra.QuasiIndex(str(d))


def test_index_open_and_metadata(sample_data):
    import rapmap_amd as ra
    from oracle import q5
    ix = ra.QuasiIndex(sample_data["idx"])
    ref = q5.load(sample_data["idx"])
    assert (ix.k, ix.n_txps, ix.text_len, ix.n_keys) == (31, 15, ref.text.size, ref.hkeys.size)
    assert ix.txp_names == ref.names and list(ix.txp_lens) == list(ref.txpLens)
    ix.close()


@pytest.mark.skipif(_have_gpu(), reason="checks the behaviour WITHOUT a GPU")
def test_no_gpu_is_a_loud_error_not_a_fallback(sample_data):
    import rapmap_amd as ra
    ix = ra.QuasiIndex(sample_data["idx"])
    with pytest.raises(ra.QmError, match="HIP|device"):
        ra.QuasiMapper(ix, 0)
