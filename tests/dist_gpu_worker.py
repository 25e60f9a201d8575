"""Worker of tests/test_distributed_gpu.py: one rank per GPU under torch.distributed.run.  Maps its contiguous shard of
the pairs on cuda:LOCAL_RANK through the C ABI, all-reduces the HitCounters over RCCL, leaves its shard's hits in the
output directory for the test process to concatenate.
usage: dist_gpu_worker.py INDEX_DIR WORK_DIR   (WORK_DIR holds a1.npy o1.npy a2.npy o2.npy)"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    idx, work = sys.argv[1], sys.argv[2]
    rank = int(os.environ["RANK"]); local = int(os.environ["LOCAL_RANK"]); world = int(os.environ["WORLD_SIZE"])
    share = os.environ.get("QMAP_TEST_SHARE_GPU") == "1"        # rehearsal on a 1-GPU box: ranks share the GPU, gloo instead of RCCL
    if share:
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if share:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    import rapmap_amd as ra
    from rapmap_amd import dist as qd
    a1 = np.load(os.path.join(work, "a1.npy")); o1 = np.load(os.path.join(work, "o1.npy"))
    a2 = np.load(os.path.join(work, "a2.npy")); o2 = np.load(os.path.join(work, "o2.npy"))
    n = len(o1) - 1
    b, e = qd.shard_bounds(n, rank, world)
    qi = ra.QuasiIndex(idx)
    mp = ra.QuasiMapper(qi, local)
    res = mp.map_pairs(a1[o1[b]:o1[e]], o1[b:e + 1] - o1[b], a2[o2[b]:o2[e]], o2[b:e + 1] - o2[b])
    tot = qd.all_reduce_counters(res.counters, device=device)
    if rank == 0:
        print("backend %s world %d" % (dist.get_backend(), dist.get_world_size()), flush=True)
    np.save(os.path.join(work, "hits_%d.npy" % rank), res.hits)
    np.save(os.path.join(work, "cnt_%d.npy" % rank), np.diff(res.hit_offsets))
    if rank == 0:
        np.save(os.path.join(work, "total.npy"), np.array([tot[k] for k in qd.COUNTER_KEYS], dtype=np.int64))
    dist.barrier()
    mp.close(); qi.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
