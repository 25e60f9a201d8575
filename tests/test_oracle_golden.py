"""The oracle (CPU restatement) against every known answer that exists for this path:
 * the SURVEY-recorded md5 of the reference's SAM on its own sample_data (config 1),
 * SAM bodies the reference's probe build wrote for tests/golden/synth_small (see make_golden.py)."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLD, load_oracle
from util import pack, sam_groups_from_gz, strip_seq

VARIANTS = {
    "default": {},
    "noStrictCheck": {"strictCheck": 0},
    "z0.9": {"quasiCov": 0.9},
    "m3": {"maxNumHits": 3},
    "noOrphans": {"noOrphans": 1},
    "noSensitive": {"sensitive": 0},
    "fuzzy": {"fuzzy": 1},
    "fuzzy_noOrphans_m3": {"fuzzy": 1, "noOrphans": 1, "maxNumHits": 3},
}


def test_sample_data_sam_md5(sample_data, oracle_mod):
    from rapmap_amd import sam
    ix, orc = load_oracle(sample_data["idx"])
    q1, o1 = pack(sample_data["reads1"]); q2, o2 = pack(sample_data["reads2"])
    res = orc.map_pairs(q1, o1, q2, o2, nthreads=2)
    assert res.counters["numReads"] == 10000
    assert res.counters["totHits"] == 14253            # 1.4253 hits / read
    body = []
    for i in range(10000):
        h = res.hits[res.hit_offsets[i]:res.hit_offsets[i + 1]]
        body.append(sam.format_pair(sample_data["names1"][i], sample_data["reads1"][i], sample_data["names2"][i],
                                    sample_data["reads2"][i], h, ix.names, ix.txpLens))
    body = "".join(body)
    assert body.count("\n") == 28506
    text = "".join(l for l in (sam.sam_header(ix.names, ix.txpLens) + body).splitlines(True) if not l.startswith("@PG"))
    want = open(os.path.join(GOLD, "sample_data", "expected_sam_body.md5")).read().strip()
    assert hashlib.md5(text.encode()).hexdigest() == want


@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_synth_small_sam(synth_small, oracle_mod, variant):
    from rapmap_amd import sam
    ix, orc = load_oracle(synth_small["idx"])
    q1, o1 = pack(synth_small["reads1"]); q2, o2 = pack(synth_small["reads2"])
    opts = oracle_mod.default_opts(**VARIANTS[variant])
    res = orc.map_pairs(q1, o1, q2, o2, opts=opts, nthreads=4)
    want = sam_groups_from_gz(os.path.join(GOLD, "synth_small", "expected_%s.noseq.sam.gz" % variant))
    bad = []
    for i in range(len(synth_small["reads1"])):
        h = res.hits[res.hit_offsets[i]:res.hit_offsets[i + 1]]
        txt = sam.format_pair(synth_small["names1"][i], synth_small["reads1"][i], synth_small["names2"][i],
                              synth_small["reads2"][i], h, ix.names, ix.txpLens, opts.maxNumHits)
        mine = strip_seq(txt)
        if mine != want[sam._read_name(synth_small["names1"][i])]:
            bad.append(i)
    assert not bad, "pairs whose SAM differs from the reference: %s" % bad[:10]


def test_threads_do_not_change_results(synth_small, oracle_mod):
    ix, orc = load_oracle(synth_small["idx"])
    q1, o1 = pack(synth_small["reads1"]); q2, o2 = pack(synth_small["reads2"])
    a = orc.map_pairs(q1, o1, q2, o2, nthreads=1)
    b = orc.map_pairs(q1, o1, q2, o2, nthreads=5)
    assert np.array_equal(a.hit_offsets, b.hit_offsets) and a.hits.tobytes() == b.hits.tobytes()
    assert a.counters == b.counters
