"""The N>1 path on CPU: two gloo ranks shard the read pairs (contiguous static split, no data-path
collective), each maps its shard, one all-reduce sums the HitCounters.  The
per-rank engine here is the oracle (there is no GPU in this container); on the GPU box the same
rapmap_amd.dist helpers wrap the HIP mapper over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp_

from conftest import ROOT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, idx, q1, o, q2, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle, q5
    from rapmap_amd import dist as qd
    n = len(o) - 1
    b, e = qd.shard_bounds(n, rank, world)
    so = o[b:e + 1] - o[b]
    orc = oracle.Oracle(q5.load(idx))
    res = orc.map_pairs(q1[o[b]:o[e]], so, q2[o[b]:o[e]], so, nthreads=2)
    tot = qd.all_reduce_counters(res.counters, device="cpu")
    np.save(os.path.join(out_dir, "hits_%d.npy" % rank), res.hits)
    np.save(os.path.join(out_dir, "cnt_%d.npy" % rank), np.diff(res.hit_offsets))
    if rank == 0:
        np.save(os.path.join(out_dir, "total.npy"), np.array([tot[k] for k in qd.COUNTER_KEYS], dtype=np.int64))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process(synth_small, oracle_mod, tmp_path):
    from oracle import oracle, q5
    from rapmap_amd import dist as qd
    from util import pack
    port = _free_port()
    # the worker slices both mates with one offsets array, so keep the pairs whose mates have equal length
    reads1 = synth_small["reads1"]; reads2 = synth_small["reads2"]
    keep = [i for i in range(len(reads1)) if len(reads1[i]) == len(reads2[i])]
    r1 = [reads1[i] for i in keep]; r2 = [reads2[i] for i in keep]
    a1, ao = pack(r1); a2, _ = pack(r2)
    ref = oracle.Oracle(q5.load(synth_small["idx"])).map_pairs(a1, ao, a2, ao, nthreads=4)
    mp_.spawn(_worker, args=(2, port, synth_small["idx"], a1, ao, a2, str(tmp_path)), nprocs=2, join=True)
    hits = np.concatenate([np.load(tmp_path / "hits_0.npy"), np.load(tmp_path / "hits_1.npy")])
    cnt = np.concatenate([np.load(tmp_path / "cnt_0.npy"), np.load(tmp_path / "cnt_1.npy")])
    assert np.array_equal(cnt, np.diff(ref.hit_offsets)) and hits.tobytes() == ref.hits.tobytes()
    tot = np.load(tmp_path / "total.npy")
    assert [int(x) for x in tot] == [ref.counters[k] for k in qd.COUNTER_KEYS]


def test_shard_bounds_cover_everything():
    from rapmap_amd import dist as qd
    for n in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 3, 8):
            b = [qd.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(e - s for s, e in b) - min(e - s for s, e in b) <= 1


def test_counters_without_a_process_group():
    """a plain one-process run: nothing to sum over, no tensor made for it"""
    from rapmap_amd import dist as qd
    c = {k: i + 1 for i, k in enumerate(qd.COUNTER_KEYS)}
    c["extra"] = 99
    assert qd.all_reduce_counters(c, device="cpu") == {k: i + 1 for i, k in enumerate(qd.COUNTER_KEYS)}
