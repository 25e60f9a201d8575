"""The N>1 path on real GPUs: mirror of tests/test_distributed_cpu.py with the HIP mapper as the per-rank engine and
RCCL ("nccl") as the backend.  Two ranks shard the read pairs (contiguous static split, no data-path collective),
each maps its shard on its own GPU through the C ABI, one all-reduce sums the HitCounters.
Needs >= 2 GPUs in the box: skipped on the 1-GPU box, runs on the driver's multi-GPU node."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _n_gpus():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def _launch(nproc, script_args, env=None):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + script_args
    e = dict(os.environ); e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if env:
        e.update(env)
    return subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=1500)


@pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs (RCCL over xGMI)")
def test_two_ranks_over_rccl_match_the_oracle(synth_small, oracle_mod, tmp_path):
    from oracle import oracle, q5
    from rapmap_amd import dist as qd
    from util import pack
    reads1 = synth_small["reads1"]; reads2 = synth_small["reads2"]
    a1, o1 = pack(reads1); a2, o2 = pack(reads2)
    np.save(tmp_path / "a1.npy", a1); np.save(tmp_path / "o1.npy", o1); np.save(tmp_path / "a2.npy", a2); np.save(tmp_path / "o2.npy", o2)
    ref = oracle.Oracle(q5.load(synth_small["idx"])).map_pairs(a1, o1, a2, o2, nthreads=4)
    r = _launch(2, [os.path.join(ROOT, "tests", "dist_gpu_worker.py"), synth_small["idx"], str(tmp_path)])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    hits = np.concatenate([np.load(tmp_path / "hits_0.npy"), np.load(tmp_path / "hits_1.npy")])
    cnt = np.concatenate([np.load(tmp_path / "cnt_0.npy"), np.load(tmp_path / "cnt_1.npy")])
    assert np.array_equal(cnt, np.diff(ref.hit_offsets)) and hits.tobytes() == ref.hits.tobytes()
    tot = np.load(tmp_path / "total.npy")
    assert [int(x) for x in tot] == [ref.counters[k] for k in qd.COUNTER_KEYS]


def test_one_rank_over_rccl_matches_the_oracle(synth_small, oracle_mod, tmp_path):
    """RCCL itself on the 1-GPU box: torch.distributed.run with ONE rank on the "nccl" backend (no gloo, no shared-GPU rehearsal) --
    the communicator is created on cuda:0 and the HitCounters go through a real all-reduce; shard = the whole input"""
    from oracle import oracle, q5
    from rapmap_amd import dist as qd
    from util import pack
    a1, o1 = pack(synth_small["reads1"]); a2, o2 = pack(synth_small["reads2"])
    np.save(tmp_path / "a1.npy", a1); np.save(tmp_path / "o1.npy", o1); np.save(tmp_path / "a2.npy", a2); np.save(tmp_path / "o2.npy", o2)
    ref = oracle.Oracle(q5.load(synth_small["idx"])).map_pairs(a1, o1, a2, o2, nthreads=4)
    r = _launch(1, [os.path.join(ROOT, "tests", "dist_gpu_worker.py"), synth_small["idx"], str(tmp_path)], env={"QMAP_TEST_SHARE_GPU": "0"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "backend nccl world 1" in r.stdout, r.stdout[-2000:]
    assert np.array_equal(np.load(tmp_path / "cnt_0.npy"), np.diff(ref.hit_offsets)) and np.load(tmp_path / "hits_0.npy").tobytes() == ref.hits.tobytes()
    assert [int(x) for x in np.load(tmp_path / "total.npy")] == [ref.counters[k] for k in qd.COUNTER_KEYS]


def test_bench_one_rank_on_the_nccl_backend(tmp_path):
    """bench.py as the driver launches it for N>1, with one rank: process group on "nccl", the counter all-reduce (one after the K timed
    steps, one untimed behind the warm-up) and the barriers around the timed region run through RCCL; the line says which backend carried them"""
    e = dict(os.environ)
    e["QMAP_BENCH_CACHE"] = str(tmp_path); e.pop("QMAP_BENCH_REHEARSAL", None)
    r = _launch(1, [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--genes", "800", "--pairs", "200000",
                    "--cpu-seconds", "2", "--no-other-configs", "--no-side-legs"], env=e)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["collective"]["backend"] == "nccl" and out["collective"]["world_size"] == 1
    assert out["collective"]["calls"] == 2 and out["collective"]["sum_equals_rank_sums"] is True
    assert out["parity"]["bit_identical_to_oracle"] is True


def _rehearse_worker(synth_small, tmp_path, nproc):
    from oracle import oracle, q5
    from rapmap_amd import dist as qd
    from util import pack
    a1, o1 = pack(synth_small["reads1"]); a2, o2 = pack(synth_small["reads2"])
    np.save(tmp_path / "a1.npy", a1); np.save(tmp_path / "o1.npy", o1); np.save(tmp_path / "a2.npy", a2); np.save(tmp_path / "o2.npy", o2)
    ref = oracle.Oracle(q5.load(synth_small["idx"])).map_pairs(a1, o1, a2, o2, nthreads=4)
    r = _launch(nproc, [os.path.join(ROOT, "tests", "dist_gpu_worker.py"), synth_small["idx"], str(tmp_path)], env={"QMAP_TEST_SHARE_GPU": "1"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    hits = np.concatenate([np.load(tmp_path / ("hits_%d.npy" % i)) for i in range(nproc)])
    cnt = np.concatenate([np.load(tmp_path / ("cnt_%d.npy" % i)) for i in range(nproc)])
    assert np.array_equal(cnt, np.diff(ref.hit_offsets)) and hits.tobytes() == ref.hits.tobytes()
    tot = np.load(tmp_path / "total.npy")
    assert [int(x) for x in tot] == [ref.counters[k] for k in qd.COUNTER_KEYS]


@pytest.mark.parametrize("nproc", [2, 3])
def test_ranks_sharing_one_gpu_match_the_oracle(synth_small, oracle_mod, tmp_path, nproc):
    """rehearsal of the N>1 path where only one GPU exists: the ranks share it and all-reduce over gloo -- everything but
    RCCL itself (shard bounds, per-rank mapping through the C ABI, counter sum, shard concatenation) against the oracle"""
    _rehearse_worker(synth_small, tmp_path, nproc)


def test_bench_two_ranks_rehearsal(tmp_path):
    """`python bench.py --gpus 2` end to end on whatever GPUs there are (QMAP_BENCH_REHEARSAL: shared GPU, gloo): the
    launcher hand-over, rank 0 building the index behind a barrier, per-rank seeds, max-over-ranks timing, ONE line from rank 0"""
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e["QMAP_BENCH_CACHE"] = str(tmp_path); e["QMAP_BENCH_REHEARSAL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--genes", "800", "--pairs", "200000"], cwd=ROOT, env=e, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["scaling"] == "weak"
    assert out["config"]["pairs_per_gpu_per_step"] == 200000


@pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs")
def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it must run two ranks and say so"""
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e["QMAP_BENCH_CACHE"] = str(tmp_path)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--genes", "800", "--pairs", "200000"], cwd=ROOT, env=e, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["scaling"] == "weak"
    assert out["config"]["pairs_per_gpu_per_step"] == 200000


def test_bench_single_gpu_line_has_the_contract_fields(tmp_path):
    """the N=1 line on a small workload: metric fields, roofline, cpu_baseline, parity against the oracle"""
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e["QMAP_BENCH_CACHE"] = str(tmp_path)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--genes", "800",
                        "--pairs", "100000", "--cpu-seconds", "2"], cwd=ROOT, env=e, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["parity"]["bit_identical_to_oracle"] is True
    assert out["roofline"]["bound"] == "hbm" and 0 < out["roofline"]["frac"] < 1
    assert out["cpu_baseline"]["kind"] in ("port", "reference") and out["cpu_baseline"]["value"] > 0
