"""Multi-GPU: one process per GPU, read pairs statically sharded, full index replica per GPU, and ONE
collective -- the sum of the HitCounters (include/RapMapUtils.hpp:208-216) -- after the last batch
(SURVEY.md section 8e).  torch.distributed is plumbing here: backend "nccl" is RCCL over xGMI on the
GPU box, "gloo" in the CPU tests."""
import torch
import torch.distributed as dist

COUNTER_KEYS = ["peHits", "seHits", "totHits", "numReads", "tooManyHits", "mappedUnits"]


def shard_bounds(n, rank, world):
    """contiguous static split: shard g = units [g*n/W, (g+1)*n/W)"""
    return (n * rank) // world, (n * (rank + 1)) // world


def all_reduce_counters(counters, device):
    """dict of the six counters -> dict of their sums over all ranks (48 bytes on the wire)"""
    if not (dist.is_available() and dist.is_initialized()):
        # no process group (a plain one-process run): nothing to sum over -- and no tensor round trip through the device for it (0.5-0.8 ms of
        # host latency per call behind a 21 ms step, profiles/r06/timeline.sh)
        return {k: int(counters[k]) for k in COUNTER_KEYS}
    t = torch.tensor([int(counters[k]) for k in COUNTER_KEYS], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)                # (also with ONE rank under torchrun: the collective library runs, the sum is the input)
    return dict(zip(COUNTER_KEYS, (int(x) for x in t.cpu())))
