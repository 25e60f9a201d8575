#!/usr/bin/env python3
"""bench.py -- whole-job throughput of the quasi-mapping hot path on N MI355X of one node.

One "step" = one pass of the hot path (qm_map_device: SACollector x2 -> hitsToMappingsSimple x2 ->
mergeLeftRightHits for every pair, then the CSR compaction of the hits and the counter read-back)
over one batch of synthetic 2x100 bp read pairs that is already resident in HBM.  Workload at N=1 =
BASELINE.json configs[1]: GENCODE-like ~200k-transcript index, 10 M pairs per GPU, hits only.
For N>1 every rank owns its own shard of pairs generated from its own seed (weak scaling: 10 M pairs per GPU; at
N=8 12.5 M per GPU = BASELINE.json configs[2], 100 M pairs over the node), the index is replicated, and the only
collective is the all-reduce of the six HitCounters per step (RCCL over xGMI).  `python bench.py --gpus N` without a
launcher re-executes itself under torch.distributed.run with N ranks.

Prints ONE JSON line on rank 0 (see the driver contract) with `roofline` and `cpu_baseline` objects.  At N=1 the same
line also carries, under `other_configs`, bounded legs of configs[3] (the -p index: BooPHF walked on the device, and the
same index expanded into the bucket table at load) and configs[4] (-s), each with its own value, kernel time, roofline
and parity against the oracle, and under `pcie_inclusive` / `end_to_end` what a caller with host buffers / FASTQ files sees.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
if "--build-only" not in sys.argv:      # (BackgroundBuilds' children build an index and exit: no torch)
    import torch  # noqa: E402
    import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

# the kernels a configuration launches for stage A (template arguments: 64-character slots, waves per SIMD the launch is
# sized for, feature flags QM_F_PH = 1, QM_F_NIP = 2, QM_F_SEL = 4, QM_F_COLLECT = 8)
KERNELS = {
    "dense": "qm_duo_kernel<dense> + qm_lean_kernel<paired, plain, dense>, one part of the batch each, in flight together (stage A: the two mates of a pair in "
             "one wavefront -- in lockstep in its two halves and merged there / one after the other --, canonical bucket table; the reads they leave "
             "go through qm_read_kernel<2,8,0>)",
    "ph_compact": "qm_lean_kernel<paired, plain, -p> (stage A: two reads per wavefront and iteration, pre-filter + BooPHF levels walked per lookup)",
    "ph_expanded": "qm_lean_kernel<paired, plain, dense> (the -p index expanded into the canonical bucket table at load)",
    "sel": "qm_lean_kernel<paired, -s, dense> (chain-scoring collector) + qm_h2m_pack_kernel (intervals -> position lists, chaining; several reads per "
           "wavefront) [+ its wide edition and qm_h2m_kernel<4> for the reads those hand on]",
}
RANDOM_SECTOR_CEILING_G = 51.0   # G random 64-byte sectors per second: profiles/r01_random_gather_roofline.txt (profiles/microbench/gather_bench.hip)
# VALU issue peak for the ksw2 recurrence (-s): 1 024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction
# (MI355X_MICROARCH.md "Wave scheduling" and "Per-instruction cycle constants": v_fma_f32 wave64 2 cyc) = 1.2288e12 wave
# instructions/s; the recurrence is 13 packed 16-bit instructions for the two cells of each of 64 lanes (qm_sel.inl)
KSW2_PEAK_G_CELLS = 1024 * 2.4e9 / 2 * (128.0 / 13.0) / 1e9


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def build_or_reuse_index(genes, seed, k, rank, world, cache_root, perfect_hash=False, paralogs=0.0, repeat_family=0, wait_only=False):
    """rank 0 builds the synthetic transcriptome + quasi-index once per box; everybody mmaps it."""
    import rapmap_amd as ra
    from rapmap_amd import synth
    tag = "g%d_s%d_k%d%s%s" % (genes, seed, k, "_ph" if perfect_hash else "", ("_par%g_rep%d" % (paralogs, repeat_family)) if (paralogs or repeat_family) else "")
    d = os.path.join(cache_root, "qmap_bench_" + tag)
    idx = os.path.join(d, "idx")
    done = os.path.join(d, "DONE")
    if rank == 0 and not os.path.exists(done) and not wait_only:
        os.makedirs(d, exist_ok=True)
        t = time.time()
        names, txps = synth.make_transcriptome(genes, seed=seed, paralog_frac=paralogs, repeat_family=repeat_family)
        fa = os.path.join(d, "txome.fa")
        synth.write_fasta(fa, names, txps)
        log("transcriptome: %d transcripts, %d bases (%.1fs)" % (len(txps), sum(x.size for x in txps), time.time() - t))
        del names, txps
        t = time.time()
        ra.build_index(fa, idx, k=k, threads=min(32, os.cpu_count() or 1), perfect_hash=perfect_hash)
        os.remove(fa)
        log("quasiindex%s%s built in %.1fs" % (" -p" if perfect_hash else "", " (paralogs %g, repeat family %d)" % (paralogs, repeat_family) if (paralogs or repeat_family) else "", time.time() - t))
        open(done, "w").write("ok\n")
    # the other ranks wait for the DONE file, not in a collective: a rank that finds the index there starts at once (no rank
    # ever sits in a RCCL barrier for the length of an index build), and the replicas of all ranks upload side by side
    t0 = time.time()
    while not os.path.exists(done):
        if time.time() - t0 > 3600:
            raise SystemExit("rank %d: no index after an hour (%s)" % (rank, done))
        time.sleep(0.25)
    return idx


class BackgroundBuilds:
    """the index builds the later legs need (the -p index of configs[3], the paralog index of the input variants), started as ONE child
    process when the run begins -- they are CPU work that used to sit between the legs (2 x 35 s of a 200 s run) -- and STOPPED
    (SIGSTOP / SIGCONT of that child: its builder threads with it) around every timed region, so that no timed step shares the
    host with them"""

    def __init__(self, specs, args):
        import subprocess
        self.ps = {}
        for spec in specs:           # one child per index: they run side by side with this process's own build of the headline's index
            cmd = [sys.executable, os.path.abspath(__file__), "--build-only", spec, "--genes", str(args.genes), "--cache", args.cache]
            self.ps[spec] = subprocess.Popen(cmd, stdout=subprocess.DEVNULL)

    def ok(self, spec):
        p = self.ps.get(spec)
        return p is not None and (p.poll() is None or p.returncode == 0)

    def quiet(self):
        import contextlib
        import signal

        @contextlib.contextmanager
        def cm():
            stopped = []
            for p in self.ps.values():
                if p.poll() is None:
                    try:
                        os.kill(p.pid, signal.SIGSTOP); stopped.append(p)
                    except OSError:
                        pass
            try:
                yield
            finally:
                for p in stopped:
                    try:
                        os.kill(p.pid, signal.SIGCONT)
                    except OSError:
                        pass
        return cm()

    def close(self):
        for p in self.ps.values():
            if p.poll() is None:
                p.wait()


def spec_args(spec):
    """'ph' | 'par<frac>,<family>' -> keyword arguments of build_or_reuse_index"""
    if spec == "ph":
        return dict(perfect_hash=True)
    a, b = spec[3:].split(",")
    return dict(paralogs=float(a), repeat_family=int(b))


def load_text_to_gpu(qi, device):
    """transcript starts / lengths and the concatenated text (rmi.seq) for the read generator"""
    text, offsets = qi.arrays()
    lens = torch.from_numpy(np.asarray(qi.txp_lens, dtype=np.int64))
    return torch.from_numpy(text).to(device), torch.from_numpy(offsets).to(device), lens.to(device)


def make_reads_gpu(text, starts, lens, n_pairs, seed, device, read_len=100, err=0.01, n_rate=0.0, chunk=1 << 20):
    """SURVEY.md section 8d generator, on the GPU: fragments N(250,25) clipped to [L,400] from transcripts of
    length >= 400, mate1 = first L bases, mate2 = reverse complement of the last L, mates swapped w.p. 0.5,
    i.i.d. substitutions."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    L = read_len
    ok = torch.nonzero(lens >= 400).flatten()
    comp = torch.zeros(256, dtype=torch.uint8, device=device)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    bases = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    code = torch.zeros(256, dtype=torch.int64, device=device)
    for i, a in enumerate(b"ACGT"):
        code[a] = i
    # 64 bytes of slack behind the last read: qm_map_device fetches reads a word at a time (include/qmap_mi355.h)
    s1 = torch.zeros(n_pairs * L + 64, dtype=torch.uint8, device=device)
    s2 = torch.zeros(n_pairs * L + 64, dtype=torch.uint8, device=device)
    ar = torch.arange(L, device=device)
    for b in range(0, n_pairs, chunk):
        e = min(n_pairs, b + chunk)
        m = e - b
        tid = ok[torch.randint(0, ok.numel(), (m,), generator=g, device=device)]
        flen = (torch.randn(m, generator=g, device=device) * 25 + 250).to(torch.int64).clamp(L, 400)
        flen = torch.minimum(flen, lens[tid])
        st = (torch.rand(m, generator=g, device=device, dtype=torch.float64) * (lens[tid] - flen + 1).double()).to(torch.int64)
        g0 = starts[tid] + st
        a = text[g0[:, None] + ar[None, :]]
        bb = comp[text[(g0 + flen - 1)[:, None] - ar[None, :]].long()]
        for r in (a, bb):
            msk = torch.rand(r.shape, generator=g, device=device) < err
            shift = torch.randint(1, 4, r.shape, generator=g, device=device)
            sub = bases[(code[r.long()] + shift) % 4]
            r[msk] = sub[msk]
            if n_rate > 0:                                   # ("also run 0 % and 0.1 % N's")
                r[torch.rand(r.shape, generator=g, device=device) < n_rate] = ord("N")
        sw = torch.rand(m, generator=g, device=device) < 0.5
        s1[b * L:e * L] = torch.where(sw[:, None], bb, a).reshape(-1)
        s2[b * L:e * L] = torch.where(sw[:, None], a, bb).reshape(-1)
    off = torch.arange(n_pairs + 1, dtype=torch.int64, device=device) * L
    return s1, s2, off


def algorithmic_bytes_per_pair(work, n, read_len):
    """B_pair = 2L + 16 n_probe + 4 n_SA + n_text + 24 n_rank + 36 n_hits with the
    counters of the reference's sequential algorithm (emitted by the oracle on the exact input)."""
    w = {k: v / n for k, v in work.items()}
    return (2 * read_len + 16 * w["n_probe"] + 4 * w["n_sa"] + w["n_text"] + 24 * w["n_rank"] + 36 * w["n_hits"]), w


def ph_walk_addend(idx_dir, n_probe):
    """a probe of the reference's -p structure costs (levels visited) x 8 + 8 (rank sample) + 4 (data_) +
    4 (SA) + 31 (text) + 1 (lens) bytes instead of the 16 of a dense find.  Levels visited: the expectation for a k-mer that
    is not in the index (the large majority of the finds), from the bit densities of the levels of this hash_info.bph."""
    from oracle import q5ph
    boo = q5ph.BooPHF(os.path.join(idx_dir, "hash_info.bph"))
    reach, levels = 1.0, 0.0
    for (size, words, ranks), dom in zip(boo.levels, boo.domains):
        dens = float(np.unpackbits(np.ascontiguousarray(words).view(np.uint8)).sum()) / max(1, dom)
        levels += reach
        reach *= (1.0 - dens)
    return n_probe * (levels * 8 + 8 + 4 + 4 + 31 + 1 - 16), levels


def pmc_traffic(key, n, genes):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this command (profiles/pmc_traffic.json, written by
    profiles/r06/make_pmc_traffic.py from the passes of profiles/r06/pmc_all.sh): a counter run cannot happen inside this process,
    so the figure is the one measured on this workload when the profile was taken -- per pair, scaled to this launch -- and says so.
    -> (bytes of the dominant kernel's launch, source, the whole entry)"""
    try:
        ent = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(key) or {}
        b = ent.get("hbm_bytes_per_launch")
        if b is None or genes != 40000:
            return None, None, {}
        return b * (n / float(ent.get("pairs_per_launch", 10_000_000))), ent.get("source", "profiles/pmc_traffic.json"), ent
    except Exception:
        return None, None, {}


def fstype_of(path):
    best, typ = "", "?"
    try:
        for line in open("/proc/mounts"):
            f = line.split()
            if len(f) >= 3 and os.path.abspath(path).startswith(f[1]) and len(f[1]) > len(best):
                best, typ = f[1], f[2]
    except Exception:
        pass
    return typ


def interleave_memory(on):
    """set_mempolicy(MPOL_INTERLEAVE over all nodes) for the allocations that follow (the oracle's 12 GB index is probed at
    random by threads on both sockets: first-touch placement would put all of it behind one socket's memory controllers)"""
    try:
        libc = ctypes.CDLL(None, use_errno=True)
        if on:
            nodes = [int(x[4:]) for x in os.listdir("/sys/devices/system/node") if x.startswith("node") and x[4:].isdigit()]
            if len(nodes) < 2:
                return False
            mask = ctypes.c_ulong(sum(1 << i for i in nodes))
            return libc.syscall(238, 3, ctypes.byref(mask), ctypes.c_ulong(max(nodes) + 2)) == 0
        return libc.syscall(238, 0, None, ctypes.c_ulong(0)) == 0
    except Exception:
        return False


class Oracles:
    """the CPU restatement, one instance per index directory (test infrastructure: the checker and the cpu_baseline leg)"""

    def __init__(self):
        self.cache = {}
        self.interleaved = False

    def get(self, idx_dir):
        if idx_dir not in self.cache:
            from oracle import oracle, q5
            t = time.time()
            oracle.build()
            self.interleaved = interleave_memory(True)
            try:
                self.cache[idx_dir] = oracle.Oracle(q5.load(idx_dir, enum_cache=os.path.join(os.path.dirname(idx_dir), "oracle_enum")))
            finally:
                interleave_memory(False)
            log("oracle index %s ready (%.1fs)" % (os.path.basename(os.path.dirname(idx_dir)), time.time() - t))
        return self.cache[idx_dir]


def timed_steps(mp, opts, ptr, n, L, steps, warmup, world, device, qd, host=None):
    """W untimed + K timed passes of the hot path, bracketed by barrier + synchronize; max over ranks.
    host = (packed1, off, exc1, packed2, exc2): --reads-from-host, the batch comes up from page-locked host memory in every step"""
    def step():
        if host is not None:
            r = mp.map_pairs_prepacked(host[0], host[1], host[2], host[3], host[1], host[4], opts=opts, fetch=False)
        else:
            r = mp.map_device(n, ptr[0], ptr[1], ptr[2], ptr[3], L, opts=opts, fetch=False)
        return r
    def add(a, b):
        return {k_: int(a.get(k_, 0)) + int(b[k_]) for k_ in b}
    for _ in range(warmup):
        r = step()
    if warmup:
        qd.all_reduce_counters(r.counters, device=device)     # (the collective library's first call sets its communicator up: not inside the timed region)
    kernel_ms = []
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mine = {}
    for _ in range(steps):
        r = step()
        kernel_ms.append(r.map_kernel_ms)
        mine = add(mine, r.counters)
    # the path's only collective (the sum of the HitCounters after the last batch): ONE all-reduce of the ranks' sums over the K steps
    tot = qd.all_reduce_counters(mine, device=device)
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    tt = torch.tensor([el], dtype=torch.float64, device=device)
    if dist.is_initialized():
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item()), kernel_ms, tot


def cpu_and_parity(orc, mp, opts, oopts, s1, s2, off, ptr, n, L, cpu_seconds, sweep=True):
    """the oracle on a bounded sample of the batch (thread sweep + sample on the same sample size when they fit the budget),
    and the HIP path's hits on exactly that sample compared bit for bit"""
    cores = os.cpu_count() or 1
    res = {}
    probe_n = min(n, 1_000_000)
    h1 = s1[: probe_n * L].cpu().numpy(); h2 = s2[: probe_n * L].cpu().numpy(); ho = off[: probe_n + 1].cpu().numpy()
    sweep_rates = {}
    cand = sorted({cores, max(1, cores // 2), max(1, cores // 4)}, reverse=True) if sweep else [max(1, cores // 4)]
    orc.map_pairs(h1[: 20000 * L], ho[:20001], h2[: 20000 * L], ho[:20001], opts=oopts, nthreads=cand[-1])   # page the index in
    for T in cand:
        r = orc.map_pairs(h1, ho, h2, ho, opts=oopts, nthreads=T)
        sweep_rates[T] = probe_n / r.map_seconds
    best_t = max(sweep_rates, key=sweep_rates.get)
    rate = sweep_rates[best_t]
    sample = int(min(n, max(probe_n, rate * cpu_seconds)))
    h1 = s1[: sample * L].cpu().numpy(); h2 = s2[: sample * L].cpu().numpy(); ho = off[: sample + 1].cpu().numpy()
    ores = orc.map_pairs(h1, ho, h2, ho, opts=oopts, nthreads=best_t)
    gr = mp.map_device(sample, ptr[0], ptr[1], ptr[2], ptr[3], L, opts=opts, fetch=True)
    parity = bool(np.array_equal(gr.hit_offsets, ores.hit_offsets) and gr.hits.tobytes() == ores.hits.tobytes())
    res.update(sample=sample, best_t=best_t, sweep={str(k_): round(v / 1e6, 4) for k_, v in sorted(sweep_rates.items())}, probe_n=probe_n,
               cpu_val=sample / ores.map_seconds / 1e6, map_seconds=ores.map_seconds, call_seconds=ores.call_seconds,
               parity=parity, hits=int(ores.hit_offsets[-1]), work=ores.work, cores=cores, ores=ores)
    return res


def roofline(bpp, w, n, kernel_ms, kernel, traffic_key, genes, step_ms=None, extra=None, whole_step=False):
    """SURVEY.md section 8d's two rooflines for the dominant kernel of a leg: (i) algorithmic bytes / kernel time against the HBM
    peak, (ii) 64-byte sectors actually fetched (TCC misses of the committed PMC pass, per pair) / kernel time against the measured
    random-sector ceiling.  whole_step (-s): the step is a chain of kernels -- bytes, traffic and time are the whole step's."""
    t_ms = step_ms if (whole_step and step_ms) else kernel_ms
    ach = bpp * n / (t_ms * 1e-3) / 1e9
    traffic, src, ent = pmc_traffic(traffic_key, n, genes)
    if whole_step and ent.get("step_hbm_bytes_per_launch") is not None:
        traffic = ent["step_hbm_bytes_per_launch"] * (n / float(ent.get("pairs_per_launch", 10_000_000)))
    out = {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
           "traffic": traffic, "traffic_source": src, "kernel": kernel, "kernel_ms": round(kernel_ms, 3),
           "algorithmic_bytes_per_pair": round(bpp, 1), "per_pair_counters": {kk: round(v, 2) for kk, v in w.items()}}
    # a batch of 2 M pairs and more is mapped as parts in flight (qm_map_device, QM_SPLIT, default 2): the parts' stage-A launches run
    # AT THE SAME TIME, each over its share of the pairs and each about as long as kernel_ms -- the HIP-event span from the first
    # launch's start to the last one's end (rocprofv3's average duration per launch of the same command:
    # profiles/r06/kernel_stats_default_two_parts_r06e.txt).  achieved = the bytes of all of them / that span: the chip's rate.
    parts = int(os.environ.get("QM_SPLIT", "2")) if n >= (1 << 21) else 1
    parts = max(1, min(8, parts))
    out["launches_per_step"] = parts
    out["launches_run_concurrently"] = parts > 1
    out["pairs_per_launch"] = n // parts
    if traffic is not None and not whole_step:
        out["traffic_per_launch"] = traffic / parts          # (`traffic` is the step's: all launches)
    out["achieved_is"] = ("algorithmic bytes of the step's %d launch(es) (%d pairs each%s) / kernel_ms, the span of stage A measured with HIP events on "
                          "the library's streams; the committed PMC passes (profiles/r06/pmc_all.sh) run the batch as ONE launch (QM_SPLIT=1) of either "
                          "stage-A kernel, so that a dispatch is the whole batch: 21.7 ms (pair kernel) / 20.2 ms (qm_lean_kernel) there" % (parts, n // parts, ", in flight together" if parts > 1 else ""))
    if whole_step:
        out["frac_is"] = "whole step: algorithmic bytes of the step / ms_per_step (kernel_ms -- stage A of the two parts in flight -- is reported next to it)"
        if step_ms:
            out["stage_a_share_of_step"] = round(kernel_ms / step_ms, 3)
    sect = ent.get("step_sectors_per_pair" if whole_step else "sectors_per_pair")
    if sect is not None and genes == 40000:
        gs = sect * n / (t_ms * 1e-3) / 1e9
        out["random_access"] = {"sectors_per_pair": round(sect, 1), "g_sectors_per_s": round(gs, 2), "ceiling_g_sectors_per_s": RANDOM_SECTOR_CEILING_G,
                                "frac": round(gs / RANDOM_SECTOR_CEILING_G, 4),
                                "source": "TCC_MISS_sum of the same committed PMC passes, per pair; ceiling: profiles/r01_random_gather_roofline.txt"}
    if step_ms:
        out["frac_of_whole_step"] = round(bpp * n / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
    if extra:
        out.update(extra)
    return out


def dp_roofline(w, pairs_per_s, step_ms, n, genes, mp=None):
    """-s: the ksw2 work as a compute roofline -- band cells of the reference's algorithm (the alignments it runs: cache misses,
    neither PERFECT nor UNGAPPED chains; oracle counters on this input) per second, against the VALU issue peak for the
    recurrence.  The device answers most of those alignments without running them (sel_side_score: a gapless path that loses
    no more than one gap, and with two mismatches the best of that path and the one-gap-run paths); the cells are the
    reference's either way, and `device` says how many alignments the device's ksw2 kernel actually ran."""
    cells = w.get("n_cells", 0.0)
    _, _, ent = pmc_traffic("sel", n, genes)
    d = {"alignments_per_pair": round(w.get("n_aln", 0), 3), "band_cells_per_pair": round(cells, 1),
         "G_cell_updates_per_s": round(cells * pairs_per_s / 1e9, 2), "bound": "valu", "peak_G_cell_updates_per_s": round(KSW2_PEAK_G_CELLS, 1),
         "frac": round(cells * pairs_per_s / 1e9 / KSW2_PEAK_G_CELLS, 5),
         "peak_is": "1024 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction x 128 cells per 13 packed 16-bit instructions (the recurrence alone)",
         "note": "whole-step rate (the alignment kernels overlap the plan kernels of the following chunks)"}
    if mp is not None:
        try:
            d["device"] = {"alignment_questions_per_pair": round(mp.stat(6) / float(n), 3), "ksw2_alignments_run_per_pair": round(mp.stat(7) / float(n), 3),
                           "strip_alignments_run_per_pair": round(mp.stat(8) / float(n), 3),
                           "what": "qm_ctx_stat of the last call: hits x mates beyond PERFECT chains that getAlnScore is asked about; those of them the "
                                   "ksw2 kernel ran; those an exact DP over the diagonals -7 .. +8 answered instead (gapless path within q + 7 e of the best "
                                   "possible); the rest: alignment-cache hits, ungapped chains, answers known without any DP"}
        except Exception as ex:  # noqa: BLE001
            d["device"] = {"error": repr(ex)}
    if ent.get("ksw2_kernel_ms_per_step") and genes == 40000:
        km = ent["ksw2_kernel_ms_per_step"] * (n / float(ent.get("pairs_per_launch", 10_000_000)))
        run = d.get("device", {}).get("ksw2_alignments_run_per_pair")
        share = min(1.0, run / w["n_aln"]) if run is not None and w.get("n_aln", 0) > 0 else 1.0      # the cells of the alignments the kernel itself ran
        d["in_ksw2_kernels"] = {"kernel_ms_per_step": round(km, 2), "share_of_the_reference_alignments_run": round(share, 4),
                                "G_cell_updates_per_s": round(share * cells * n / (km * 1e-3) / 1e9, 1),
                                "frac": round(share * cells * n / (km * 1e-3) / 1e9 / KSW2_PEAK_G_CELLS, 4),
                                "source": ent.get("source", "profiles/pmc_traffic.json") + " (kernel-trace sum of the alignment kernels of one step)"}
    return d


def build_compat_bench(out_dir):
    """tests/compat/compat_bench.cpp (a caller written against the reference's call sequence, include/qmap_rapmap_compat.hpp alone)
    -> executable; g++ only"""
    import subprocess
    import rapmap_amd
    exe = os.path.join(out_dir, "compat_bench")
    src = os.path.join(ROOT, "tests", "compat", "compat_bench.cpp")
    lib_path = rapmap_amd.LIB_PATH
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-o", exe, lib_path,
                           "-Wl,-rpath," + os.path.dirname(lib_path), "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-pthread"])
    return exe


def compat_digest(hit_offsets, hits):
    """the digest compat_bench.cpp prints (hit_digest there), from CSR hits: per hit a 64-bit mix of (unit, index in unit, tid,
    pos, matePos, orientation/status flags, fragLen), summed with wrap-around"""
    off = np.asarray(hit_offsets, dtype=np.int64)
    n = len(off) - 1
    cnt = np.diff(off)
    unit = np.repeat(np.arange(n, dtype=np.uint64), cnt)
    j = (np.arange(int(off[-1]), dtype=np.int64) - np.repeat(off[:-1], cnt)).astype(np.uint64)
    h = hits[: int(off[-1])]
    u64 = lambda a: np.asarray(a).astype(np.uint64)
    paired = h["mate_status"] == 3
    mate_pos = np.where(paired, h["mate_pos"].astype(np.int64) & 0xFFFFFFFF, 0).astype(np.uint64)
    frag = np.where(paired, h["frag_len"], 0).astype(np.uint64)
    flags = (h["fwd"] != 0).astype(np.uint64) | (np.where(paired, h["mate_is_fwd"] != 0, True).astype(np.uint64) << np.uint64(1)) | (u64(h["mate_status"]) << np.uint64(2))
    K = [np.uint64(x) for x in (0x9E3779B97F4A7C15, 0xD6E8FEB86659FD93, 0xC2B2AE3D27D4EB4F, 0x165667B19E3779F9, 0x27D4EB2F165667C5,
                                 0x85EBCA77C2B2AE63, 0xFF51AFD7ED558CCD, 0xC4CEB9FE1A85EC53)]
    with np.errstate(over="ignore"):
        v = unit * K[0] + j * K[1] + u64(h["tid"]) * K[2] + (h["pos"].astype(np.int64) & 0xFFFFFFFF).astype(np.uint64) * K[3] + mate_pos * K[4] + flags * K[5] + frag * K[6]
        v ^= v >> np.uint64(31); v *= K[7]; v ^= v >> np.uint64(29)
        return "%016x" % int(v.sum(dtype=np.uint64))


def compat_face_leg(out, args, idx_dir, hs1, hs2, n, L, want_digest_fn, cores):
    """the reference's own call surface (SACollector::operator() / hitsToMappingsSimple / mergeLeftRightHits per read, read groups
    of 10 000 pairs as the parser hands them out, src/RapMapSAMapper.cpp:853) at 1 / 8 / 32 host threads"""
    import subprocess
    import tempfile
    nc = int(min(n, args.compat_pairs))
    # (the executable cannot live under /dev/shm: mounted noexec on the GPU box)
    with tempfile.TemporaryDirectory(dir=args.e2e_dir if os.path.isdir(args.e2e_dir) else None) as td:
        exe = build_compat_bench(td)
        rp = os.path.join(td, "reads.bin")
        with open(rp, "wb") as f:
            f.write(hs1[: nc * L].tobytes()); f.write(hs2[: nc * L].tobytes())
        def sweep(extra):
            runs = {}
            for T in sorted({1, min(8, cores), min(32, cores)}):
                npairs = nc if T > 1 else min(nc, max(200_000, nc // 8))          # one thread: an eighth of the sample is plenty
                r = subprocess.run([exe, idx_dir, rp, str(nc), str(L), str(T), "10000", "--use", str(npairs)] + ["--repeat", "3" if T > 1 else "4", "--digest-once"] + extra,     # best of the repeats behind the first, which also page-locks the service's buffers and takes the digest
                                   capture_output=True, text=True, timeout=900)
                line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                if r.returncode != 0 or not line:
                    runs[str(T)] = {"error": (r.stdout + r.stderr)[-300:]}
                    continue
                j = json.loads(line[-1])
                runs[str(T)] = {"value": round(j["mpairs_per_s"], 3), "pairs": j["pairs"], "seconds": round(j["seconds"], 4),
                                "thread_join_seconds_not_timed": round(j.get("thread_join_seconds", 0.0), 4),
                                "workers_waiting_thread_s": round(j.get("prefetch_thread_s", 0.0), 3), "workers_loop_thread_s": round(j.get("loop_thread_s", 0.0), 3),
                                "bit_identical_joint_hits": j["digest"] == want_digest_fn(npairs), "totHits": j["totHits"]}
            return runs
        runs = sweep([])
        # the same callers after ONE more line, `hitCollector.setKeepIntervals(false)` ("I hand HitCollectorInfo on and never look inside": what
        # src/RapMapSAMapper.cpp:466-486 does): no interval records come down, the device pass runs on the pair / lean kernels
        runs_ni = sweep(["--no-intervals"])
        best = max((v.get("value", 0) for v in runs.values()), default=0)
        out["compat_face"] = {"value": best, "unit": "M read-pairs/s", "chunk_pairs": 10000, "by_host_threads": runs,
                              "without_interval_records": {"by_host_threads": runs_ni, "value": max((v.get("value", 0) for v in runs_ni.values()), default=0),
                                                           "what": "the same run with SACollector::setKeepIntervals(false): fwdSAInts / rcSAInts of every HitCollectorInfo stay empty, "
                                                                   "everything else -- foundHit, hit lists, merges -- unchanged (same digest)"},
                              "bytes_per_pair": {"up": "56 (2-bit packed reads + offsets; 216 as characters, QMAP_COMPAT_NO_PACK=1)",
                                                 "down": "about 330 with the interval records, 170 without (hits 32 B x 3.2, list words 8 B x 6.4, offsets, flags)"},
                              "what": "tests/compat/compat_bench.cpp: T threads, each takes read groups of 10 000 pairs and runs the reference's per-pair "
                                      "sequence (src/RapMapSAMapper.cpp:461-551) through include/qmap_rapmap_compat.hpp, one added line "
                                      "`hitCollector.prefetch(rg)` (the groups of all threads are mapped together by the header's batching service: "
                                      "two dispatcher threads own the device contexts); strings built before the timed region, which ends when the last "
                                      "group has been processed (joining the worker threads -- milliseconds of MMU-notifier work per exiting thread in a "
                                      "process with GPU mappings -- is reported next to it); digest of every jointHits vector against the fused path's "
                                      "hits on the same pairs (themselves checked against the oracle in `parity`), taken in the first repeat; the timed "
                                      "repeats run the reference's loop without the digest (--digest-once), their hit counters still reported"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--genes", type=int, default=40000, help="synthetic genes (40000 ~ 200k transcripts, 3e8 bases)")
    ap.add_argument("--pairs", type=int, default=None, help="read pairs per GPU per step (default: 10 M = configs[1]; "
                    "12.5 M at 8 GPUs = configs[2], 100 M pairs over the node)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="N=1: skip the bounded legs of configs[3] and configs[4]")
    ap.add_argument("--no-side-legs", action="store_true", help="N=1: skip pcie_inclusive / end_to_end")
    ap.add_argument("--sel-aln", action="store_true", help="config 5: selective alignment (-s): chaining + ksw2 extension alignment of every hit")
    ap.add_argument("--perfect-hash", action="store_true", help="config 4: index built with `quasiindex -p` (BooPHF / FrugalBooMap probe path)")
    ap.add_argument("--ph-compact", action="store_true", help="with --perfect-hash: keep the BooPHF / FrugalBooMap structure on the device "
                    "(walked per lookup) instead of expanding the -p index into the one-sector bucket table at load time")
    ap.add_argument("--err", type=float, default=0.01, help="substitution rate of the simulated reads (SURVEY.md 8d: 1 %%; the input_variants legs also run 0)")
    ap.add_argument("--n-rate", type=float, default=0.0, help="share of the read characters replaced by N (the input_variants legs run 0.1 %%)")
    ap.add_argument("--paralogs", type=float, default=0.0, help="that share of the transcripts again as paralogs with 4 %% substitutions (the input_variants legs run 5 %%)")
    ap.add_argument("--repeat-family", type=int, default=0, help="one family of that many transcripts around a shared 300-base core (the input_variants legs: 100)")
    ap.add_argument("--no-input-variants", action="store_true", help="N=1: skip the bounded legs on other input distributions (0 %% errors; N's; paralogs + a repeat family)")
    ap.add_argument("--variant-pairs", type=int, default=4_000_000, help="pairs per input_variants leg (parity on the first 2 M of them)")
    ap.add_argument("--reads-from-host", action="store_true", help="every step takes the batch from page-locked HOST buffers, 2-bit packed (qm_map_pairs_packed: upload inside the "
                    "timed region, as `quasimap --devices` feeds its GPUs) instead of reads resident in HBM: what a scaling run of the PRODUCT path would see; "
                    "never the contract's `value` (the line says so in config.reads_from_host)")
    ap.add_argument("--build-only", default=None, help=argparse.SUPPRESS)      # (BackgroundBuilds' child: build these indices and exit)
    ap.add_argument("--read-len", type=int, default=100, help="read length (BASELINE.json: 100; 129..256 runs the NS=4 kernels)")
    ap.add_argument("--cache", default=os.environ.get("QMAP_BENCH_CACHE", "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"))
    ap.add_argument("--e2e-dir", default=os.environ.get("QMAP_BENCH_E2E_DIR", "/tmp"), help="where the end_to_end leg puts its FASTQ files")
    ap.add_argument("--e2e-threads", type=int, default=24, help="ingest workers of the end_to_end leg (the engine alone peaks at 16-32 on this host, pinned to the socket that holds the files: profiles/r04/ingest_after_pin.log)")
    ap.add_argument("--gz-leg", type=int, default=0, help="N=1: also map the first GZ_LEG pairs from ordinary `gzip -6` files (end_to_end_gzip; opt-in: writing the files takes about a minute per 4 M pairs)")
    ap.add_argument("--e2e-copies", type=int, default=4, help="the end_to_end leg's FASTQ files hold the batch this many times (4 x 10 M = 40 M pairs)")
    ap.add_argument("--compat-pairs", type=int, default=8_000_000, help="pairs of the batch the compat_face leg runs through the reference's call surface")
    args = ap.parse_args()

    if args.build_only:
        os.nice(5)
        for spec in args.build_only.split(";"):
            idx = build_or_reuse_index(args.genes, 42, 31, 0, 1, args.cache, **spec_args(spec))
            if spec == "ph":
                # ... and the table the ORACLE of that index needs (the k-mers of a -p index enumerated from its suffix array: 20 s of numpy that sat
                # between two legs of the parent) -- test infrastructure prepared in the background like the index itself
                from oracle import q5
                q5.load(idx, enum_cache=os.path.join(os.path.dirname(idx), "oracle_enum"))
                open(os.path.join(os.path.dirname(idx), "DONE_ORACLE"), "w").write("ok\n")
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as a plain `python bench.py --gpus N`: become the launcher the driver would have used -- one rank per GPU
        # under torch.distributed.run, rendezvous on 127.0.0.1 -- and hand its exit code on.  Rank 0's JSON line goes to our stdout.
        import socket
        import subprocess
        s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and rank == 0:
        log("--gpus %d but WORLD_SIZE=%d: the launcher decides, running %d rank(s)" % (args.gpus, world, world))
    if args.pairs is None:
        args.pairs = 12_500_000 if world == 8 else 10_000_000
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # QMAP_BENCH_REHEARSAL=1 (tests only): the ranks share the visible GPUs and talk over gloo, so that everything of the
    # N>1 path but RCCL itself -- launcher hand-over, index build by rank 0 behind a barrier, shard seeds, counter
    # all-reduce, max-over-ranks timing, rank 0's line -- runs on a 1-GPU box.  Its numbers mean nothing.
    rehearsal = os.environ.get("QMAP_BENCH_REHEARSAL") == "1"
    if local_rank >= torch.cuda.device_count() and not rehearsal:
        raise SystemExit("rank %d has no GPU of its own (%d visible): one process per GPU" % (local_rank, torch.cuda.device_count()))
    dev_id = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    torch.cuda.set_device(dev_id)
    device = torch.device("cuda", dev_id)
    # a process group whenever a launcher set one up -- also for ONE rank (torch.distributed.run --nproc-per-node 1, or RANK /
    # WORLD_SIZE / MASTER_* in the environment): RCCL then creates its communicator and runs the per-step all-reduce on a single
    # GPU, which is the only way that code executes on a 1-GPU box
    grouped = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    import rapmap_amd as ra
    from rapmap_amd import dist as qd

    k, L = 31, args.read_len
    plain_head = not args.perfect_hash and not args.sel_aln and L == 100 and not args.paralogs and not args.repeat_family
    bg_specs = []
    if rank == 0 and world == 1 and not args.no_cpu_baseline and plain_head:
        if not args.no_other_configs:
            bg_specs.append("ph")
        if not args.no_input_variants:
            bg_specs.append("par0.05,100")
    bg = BackgroundBuilds(bg_specs, args)
    args._bg = bg
    idx_dir = build_or_reuse_index(args.genes, 42, k, rank, world, args.cache, args.perfect_hash, paralogs=args.paralogs, repeat_family=args.repeat_family)
    args._idx_dir = idx_dir
    t = time.time()
    qi = ra.QuasiIndex(idx_dir)
    mp = ra.QuasiMapper(qi, dev_id, ph_compact=args.ph_compact, wide_reads=128 < L <= 256)      # (QM_CTX_WIDE_READS: the wide extension table with the replica, not inside the first timed call)
    if rank == 0:
        log("index in HBM: %d transcripts, %d text bytes, %d k-mers, %.2f GB on device (%.1fs)" % (
            qi.n_txps, qi.text_len, qi.n_keys, mp.device_bytes / 1e9, time.time() - t))
    text, starts, lens = load_text_to_gpu(qi, device)
    s1, s2, off = make_reads_gpu(text, starts, lens, args.pairs, 43 + rank, device, read_len=L, err=args.err, n_rate=args.n_rate)
    torch.cuda.synchronize()
    n = args.pairs
    opts = ra.default_opts(sel_aln=1) if args.sel_aln else ra.default_opts()
    oopts_kw = {"selAln": 1} if args.sel_aln else {}
    ptr = (s1.data_ptr(), off.data_ptr(), s2.data_ptr(), off.data_ptr())
    head_key = "sel" if args.sel_aln else (("ph_compact" if args.ph_compact else "ph_expanded") if args.perfect_hash else "dense")

    host = None
    if args.reads_from_host:
        # the batch as `quasimap --devices` holds it: 2-bit packed (26 bytes per 100-bp read), page-locked; every step uploads it
        hoff = off.cpu().numpy()
        p1 = ra.api.pack_2bit(s1[: n * L].cpu().numpy(), hoff); p2 = ra.api.pack_2bit(s2[: n * L].cpu().numpy(), hoff)
        host = (ra.api.pinned_copy(p1[0]), ra.api.pinned_copy(hoff), ra.api.pinned_copy(p1[2]) if len(p1[2]) else p1[2], ra.api.pinned_copy(p2[0]),
                ra.api.pinned_copy(p2[2]) if len(p2[2]) else p2[2])
    with bg.quiet():
        el, kernel_ms, tot = timed_steps(mp, opts, ptr, n, L, args.steps, args.warmup, world, device, qd, host=host)
    head_stats = kernel_stats(mp, n)
    total_pairs = n * world * args.steps
    value = total_pairs / el / 1e6

    out = None
    if rank == 0:
        out = {
            "metric": "Mreads/s (paired 2x100 bp; one unit = one read pair)", "value": round(value, 4),
            "unit": "M read-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(el / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/int32/u64 (integer & byte work, no floating point)", "data": "synthetic",
            "config": {"workload": ("configs[%d]: GENCODE-like synthetic index (%d genes -> %d transcripts, %d text bytes, "
                                    "%d 31-mers), %d pairs 2x%d bp per GPU per step, 1%% substitutions, %s, %s index") % (
                                        4 if args.sel_aln else (3 if args.perfect_hash else (2 if world == 8 else 1)), args.genes, qi.n_txps, qi.text_len, qi.n_keys, n, L,
                                        "selective alignment (-s)" if args.sel_aln else "hits only (no -s)",
                                        ("perfect-hash (-p), BooPHF walked on the device" if args.ph_compact else "perfect-hash (-p), expanded into the bucket table at load") if args.perfect_hash else "dense hash"),
                       "pairs_per_gpu_per_step": n, "parallelism": "shard%d (index replicated, counters all-reduced)" % world,
                       "hits_per_pair": round(tot["totHits"] / max(1, tot["numReads"]), 4),
                       "mreads_per_s": round(2 * value, 4)},
        }
        avg_kernel_ms = float(np.mean(kernel_ms))
        out["config"]["map_kernel_ms"] = round(avg_kernel_ms, 3)
        if args.reads_from_host:
            out["config"]["reads_from_host"] = ("every step uploads the batch from page-locked host memory, 2-bit packed (%d MB per step), through qm_map_pairs_packed: the product path's "
                                                "feed, NOT the contract's HBM-resident figure" % ((host[0].nbytes + host[3].nbytes + host[1].nbytes) >> 20))
        out["config"]["reads"] = {"substitution_rate": args.err, "n_rate": args.n_rate, "paralog_share": args.paralogs, "repeat_family": args.repeat_family}
        out["stage_a_kernels"] = head_stats
        out["collective"] = ({"backend": dist.get_backend(), "world_size": dist.get_world_size(), "op": "all_reduce(SUM) of the six HitCounters, 48 bytes",
                              "calls": 1 + (1 if args.warmup else 0), "when": "one after the K timed steps (the ranks' sums over the steps), one untimed after the warm-up",
                              "sum_equals_rank_sums": bool(tot["numReads"] == n * world * args.steps)}
                             if grouped else None)

    # ---- cpu_baseline + roofline counters: rank 0, N=1 only, bounded sample of the same workload
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle
        oracles = Oracles()
        orc = oracles.get(idx_dir)
        cores = os.cpu_count() or 1
        oopts = oracle.default_opts(**oopts_kw)
        with bg.quiet():
            cp = cpu_and_parity(orc, mp, opts, oopts, s1, s2, off, ptr, n, L, args.cpu_seconds)
        sample, best_t, cpu_val = cp["sample"], cp["best_t"], cp["cpu_val"]
        bpp, w = algorithmic_bytes_per_pair(cp["work"], sample, L)
        ph_levels = None
        if args.perfect_hash and args.ph_compact:      # the expanded image never walks the levels: its bytes are the dense ones
            add, ph_levels = ph_walk_addend(idx_dir, w["n_probe"])
            bpp += add
        por = None
        try:
            por = json.load(open(os.path.join(ROOT, "profiles", "port_over_reference.json")))
        except Exception:
            pass
        out["cpu_baseline"] = {"value": round(cpu_val, 5), "unit": "M read-pairs/s", "cores": best_t, "kind": "port",
                               "host_hw_threads": cores, "threads_best": best_t,
                               "threads_sweep_Mpairs_s": cp["sweep"], "threads_sweep_pairs": cp["probe_n"],
                               "pairs_per_s_per_thread": round(sample / cp["map_seconds"] / best_t, 1),
                               "timed": "the mapping section of the port (worker threads started .. joined, %.2f s): the span the reference's own "
                                        "timer covers (src/RapMapSAMapper.cpp:856-889, output discarded with -n); the whole native call incl. "
                                        "assembling one result array took %.2f s" % (cp["map_seconds"], cp["call_seconds"]),
                               "index_memory_interleaved_over_numa_nodes": oracles.interleaved,
                               "port_over_reference": (por or {}).get("port_over_reference"),
                               "port_over_reference_note": (por or {}).get("note"),
                               "sample": "first %d pairs of the same batch, oracle (CPU restatement of the reference's algorithm) on %d threads "
                                         "(best of a sweep over %s on the first %d pairs); the reference's quasimap cannot be built in this image "
                                         "(un-vendored cereal); port / reference throughput measured in the build container: profiles/port_over_reference.json"
                                         % (sample, best_t, sorted(int(x) for x in cp["sweep"]), cp["probe_n"])}
        out["parity"] = {"sample_pairs": sample, "bit_identical_to_oracle": cp["parity"], "hits": cp["hits"]}
        extra = {}
        if ph_levels is not None:
            extra["ph_levels_per_probe"] = round(ph_levels, 3)
        if args.sel_aln:   # with -s report the DP cells separately
            extra["dp"] = dp_roofline(w, value * 1e6, el / args.steps * 1e3, n, args.genes, mp)
        klabel = KERNELS[head_key]
        if 128 < L <= 256 and not args.perfect_hash:     # the lean kernel's wide edition
            klabel = klabel.replace("two reads per wavefront and iteration", "wide edition: ONE read of up to 256 characters per wavefront and iteration, 224-character extension table")
        out["roofline"] = roofline(bpp, w, n, avg_kernel_ms, klabel, head_key, args.genes, step_ms=el / args.steps * 1e3, extra=extra,
                                   whole_step=bool(args.sel_aln))
        out["speedup_vs_cpu_baseline"] = round(value / cpu_val, 2) if cpu_val > 0 else None
        try:   # the per-pair counters are a property of the input distribution: keep them for the N>1 runs
            json.dump({"bpp": bpp, "counters": w}, open(os.path.join(idx_dir, "algorithmic_bytes.json"), "w"))
        except Exception:
            pass

        # ---- what a caller sees beyond the in-HBM figure (last bullet); never `value`
        if not args.no_side_legs:
            try:
                with bg.quiet():
                    side_legs(out, args, ra, qi, mp, opts, s1, s2, off, n, L, dev_id, cores)
            except Exception as ex:           # side measurements: they must not take the bench line down
                out.setdefault("end_to_end", None); out.setdefault("pcie_inclusive", None)
                log("side measurements failed: %r" % (ex,))

        # ---- configs[3] and configs[4] in the same run: bounded legs, each with its own value / kernel time / roofline / parity
        if not args.no_other_configs and head_key == "dense" and L == 100:
            out["other_configs"] = {}
            try:
                other_configs(out["other_configs"], args, ra, qd, oracles, oracle, qi, idx_dir, mp, s1, s2, off, ptr, n, L, dev_id, device, k, w, bpp)
            except Exception as ex:
                out["other_configs"]["error"] = repr(ex)
                log("other_configs failed: %r" % (ex,))
        # ---- other input distributions ("also run 0 % and 0.1 % N's"; paralogs): the stage-A kernels of the headline LEAVE every read they
        # are not built for to the general kernel, so how fast a batch maps depends on what is in it -- bounded legs, each with parity and its share of left reads
        if not args.no_input_variants and plain_head:
            try:
                out["input_variants"] = input_variant_legs(args, ra, qd, oracles, oracle, qi, idx_dir, mp, text, starts, lens, n, L, dev_id, device, k, value, bg)
            except Exception as ex:
                out["input_variants"] = {"error": repr(ex)}
                log("input_variants failed: %r" % (ex,))
        bg.close()
    elif rank == 0:
        # N>1 (or --no-cpu-baseline): no oracle leg.  The roofline of rank 0's kernel still uses the algorithmic bytes
        # per pair of this workload: recorded by an N=1 run on this box, else the committed figure of the default workload.
        out["cpu_baseline"] = None
        out["roofline"] = None
        default_workload = args.genes == 40000 and L == 100 and not args.perfect_hash and not args.sel_aln   # bytes per PAIR do not depend on n
        rec = None
        cands = [os.path.join(idx_dir, "algorithmic_bytes.json")]
        if default_workload:
            cands.append(os.path.join(ROOT, "profiles", "algorithmic_bytes.json"))
        for cand in cands:
            if os.path.exists(cand):
                try:
                    rec = json.load(open(cand)); break
                except Exception:
                    rec = None
        if rec and kernel_ms:
            out["roofline"] = roofline(float(rec["bpp"]), rec.get("counters") or {}, n, avg_kernel_ms,
                                       KERNELS[head_key] + ", rank 0's launches", head_key, args.genes, step_ms=el / args.steps * 1e3)

    if rank == 0:
        print(json.dumps(out), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


DEFER_WHY = ["a character that is not A C G T (an N ...) or more than 128 characters", "a window of k equal bases",
             "an SA interval wider than the kernel's lanes / more suffixes or intervals than its stash / a match beyond its extension table",
             "k-mers of the other orientation on the way: the reference maps the other strand as well"]


def kernel_stats(mp, n):
    """qm_ctx_stat of the call just made: which stage-A kernels it went through and what they left to the general kernel"""
    try:
        lean_reads, deferred = mp.stat(3), mp.stat(4)
        why = [mp.stat(11 + i) for i in range(4)]
        d = {"reads_through_the_pair_or_lean_kernel": lean_reads, "reads_left_to_the_general_kernel": deferred,
             "lean_deferred_frac": round(deferred / float(lean_reads), 6) if lean_reads > 0 else None,
             "pairs_merged_in_the_wavefront": mp.stat(10),
             "reads_with_N_mapped_by_the_N_aware_pass": mp.stat(15),
             "left_because": {DEFER_WHY[i]: why[i] for i in range(4) if why[i]},
             "what": "a batch of 2 M pairs and more is mapped as two parts in flight: one on the pair kernel (qm_duo_kernel: both mates of a pair walked in "
                     "lockstep by the two halves of a wavefront, the pair merged there), one on qm_lean_kernel (two mates per wavefront, one after the "
                     "other) -- vector-bound and scalar-bound wavefronts sharing every CU; a read neither is built for is marked and mapped by "
                     "qm_read_kernel<2,8,0> in a second, small launch inside map_kernel_ms; when 2 048 reads and more were marked for a character "
                     "outside A C G T, qm_lean_kernel's N-aware edition (k-mers with an N stepped over, MMPs ending at one) goes over the marked reads first"}
        return d
    except Exception as ex:  # noqa: BLE001
        return {"error": repr(ex)}


def input_variant_legs(args, ra, qd, oracles, oracle, qi, idx_dir, mp, text, starts, lens, n, L, dev_id, device, k, headline, bg):
    iv = {}
    m = int(min(n, args.variant_pairs))
    par = min(m, 2_000_000)
    opts = ra.default_opts()
    oopts = oracle.default_opts()
    cores = os.cpu_count() or 1

    def one(name, mapper, orc, tx, st, ln, what, seed, **gen):
        s1, s2, off = make_reads_gpu(tx, st, ln, m, seed, device, read_len=L, **gen)
        torch.cuda.synchronize()
        ptr = (s1.data_ptr(), off.data_ptr(), s2.data_ptr(), off.data_ptr())
        with bg.quiet():
            el, kms, tot = timed_steps(mapper, opts, ptr, m, L, 3, 1, 1, device, qd)
        val = m * 3 / el / 1e6
        ks = kernel_stats(mapper, m)
        h1 = s1[: par * L].cpu().numpy(); h2 = s2[: par * L].cpu().numpy(); ho = off[: par + 1].cpu().numpy()
        ores = orc.map_pairs(h1, ho, h2, ho, opts=oopts, nthreads=max(1, cores // 2))
        gr = mapper.map_device(par, ptr[0], ptr[1], ptr[2], ptr[3], L, opts=opts, fetch=True)
        ok = bool(np.array_equal(gr.hit_offsets, ores.hit_offsets) and gr.hits.tobytes() == ores.hits.tobytes() and gr.counters == ores.counters)
        leg = {"reads": what, "pairs_per_step": m, "value": round(val, 3), "unit": "M read-pairs/s", "steps": 3, "warmup": 1, "ms_per_step": round(el / 3 * 1e3, 3),
               "kernel_ms": round(float(np.mean(kms)), 3), "of_headline": round(val / headline, 3), "hits_per_pair": round(tot["totHits"] / max(1, tot["numReads"]), 4),
               "lean_deferred_frac": ks.get("lean_deferred_frac"), "stage_a_kernels": {kk: ks[kk] for kk in ks if kk != "what"},
               "parity": {"sample_pairs": par, "bit_identical_to_oracle": ok, "hits": int(ores.hit_offsets[-1])}}
        if val < 0.7 * headline:
            lb = ks.get("left_because") or {}
            top = max(lb, key=lb.get) if lb else None
            leg["below_70_percent_because"] = ("%.1f %% of the reads are left to the general kernel (one read per wavefront, 255 M pairs/s on clean reads), most of them for: %s (%d reads)"
                                               % (100.0 * (ks.get("lean_deferred_frac") or 0), top, lb[top])) if top else "no read was left: see kernel_ms against ms_per_step"
        iv[name] = leg
        log("input_variants %s: %.1f M pairs/s (%.2f of the headline), %.3f of the reads left to the general kernel, parity %s" % (name, val, val / headline, ks.get("lean_deferred_frac") or 0, ok))
        del s1, s2, off

    orc = oracles.get(idx_dir)
    one("err1 (the headline's distribution at this batch size)", mp, orc, text, starts, lens, "1 % substitutions", 143, err=0.01)
    one("err0", mp, orc, text, starts, lens, "no substitutions: every read matches its transcript end to end", 144, err=0.0)
    one("err1_N0.1", mp, orc, text, starts, lens, "1 % substitutions and 0.1 % of the characters N (9.5 % of the reads hold one)", 145, err=0.01, n_rate=0.001)
    # paralogs + a repeat family: an index of its own (built in the background since the run began)
    kw = dict(paralogs=0.05, repeat_family=100)
    idx_par = build_or_reuse_index(args.genes, 42, k, 0, 1, args.cache, wait_only=bg.ok("par0.05,100"), **kw)
    qi_p = ra.QuasiIndex(idx_par)
    mp_p = ra.QuasiMapper(qi_p, dev_id)
    tx, st, ln = load_text_to_gpu(qi_p, device)
    one("paralogs5_family100", mp_p, oracles.get(idx_par), tx, st, ln,
        "1 % substitutions; the transcriptome holds 5 % of its transcripts a second time as paralogs (4 % substitutions) and one family of 100 transcripts around a shared 300-base core: {} transcripts".format(qi_p.n_txps),
        146, err=0.01)
    mp_p.close(); qi_p.close()
    return iv


def side_legs(out, args, ra, qi, mp, opts, s1, s2, off, n, L, dev_id, cores):
    os.environ.setdefault("QM_INGEST_PIN", "1")   # the end_to_end leg's one ingest engine: workers on the NUMA node that holds the files' pages (opt-in since round 5)
    # (1) PCIe inclusive: the same batch from pageable host buffers through qm_map_pairs + qm_fetch_hits
    hs1 = s1.cpu().numpy(); hs2 = s2.cpu().numpy(); hoff = off.cpu().numpy()
    mp.map_pairs(hs1[: 1000 * L], hoff[:1001], hs2[: 1000 * L], hoff[:1001], opts=opts)
    # the context's FIRST call of this size also allocates its device buffers (2 GB of characters, lists, hits: 0.1-0.5 s by the box -- what the leg
    # reported until round 6, 15-88 M pairs/s over the boxes); a caller that maps batch after batch sees the second figure
    t = time.perf_counter(); rh = mp.map_pairs(hs1, hoff, hs2, hoff, opts=opts); dt_first = time.perf_counter() - t
    del rh                                                   # (its arrays go back to the OS before the clock starts: the next call's result lands in fresh pages again)
    t = time.perf_counter(); rh = mp.map_pairs(hs1, hoff, hs2, hoff, opts=opts); dt_h = time.perf_counter() - t
    out["pcie_inclusive"] = {"value": round(n / dt_h / 1e6, 3), "unit": "M read-pairs/s", "first_call_of_the_context": round(n / dt_first / 1e6, 3),
                             "what": "qm_map_pairs on pageable host buffers (%d MB in) + qm_fetch_hits into a fresh array (%d MB out), one call, the "
                                     "context's second of this size (first_call_of_the_context: the one before it, which also allocated the context's device buffers)"
                                     % ((2 * hs1.nbytes + 2 * hoff.nbytes) >> 20, (rh.hits.nbytes + rh.hit_offsets.nbytes) >> 20)}
    # (1b) the reference's call surface on the same pairs
    try:
        hoffs, hhits = rh.hit_offsets, rh.hits
        compat_face_leg(out, args, qi.path if hasattr(qi, "path") else args._idx_dir, hs1, hs2, n, L,
                        lambda m: compat_digest(hoffs[: m + 1], hhits), cores)
        log("compat_face: %s" % json.dumps(out["compat_face"]["by_host_threads"]))
    except Exception as ex:
        out["compat_face"] = {"error": repr(ex)}
        log("compat_face failed: %r" % (ex,))
    del rh
    # (2) end to end: two FASTQ files -> hits in pinned memory through the pipelined stream (ingest workers + device contexts).
    # The files go to a local filesystem directory by default, not to tmpfs: on this box's kernel the FIRST read of freshly
    # written tmpfs pages runs at ~16 GB/s whatever the thread count or access method (profiles/r03_tmpfs_read_passes.txt),
    # a property of the host, not of the reader; page-cache pages of a disk filesystem do not show it.
    from rapmap_amd import synth as _syn
    # the slots' pinned memory comes out of the library's process-wide pool, reserved here -- in the background, while the
    # FASTQ files are being written -- as the CLI reserves it while the index uploads (pinning runs at 5.5 GB/s on this host
    # whatever the thread count: inside the timed region the first batches would wait for their slots)
    ra.reserve_stream_memory(1280 << 20)
    # the files hold the batch's pairs `e2e_copies` times over (40 M pairs by default: at ~100 M pairs/s a 10 M-pair run is a
    # tenth of a second, inside the noise of thread start-up)
    reps = max(1, int(args.e2e_copies))
    ne = n * reps
    base = args.e2e_dir if os.path.isdir(args.e2e_dir) else args.cache
    d_e = os.path.join(base, "qmap_bench_e2e_%d" % os.getpid()); os.makedirs(d_e, exist_ok=True)
    f1 = os.path.join(d_e, "r1.fq"); f2 = os.path.join(d_e, "r2.fq")
    try:
        for rep in range(reps):
            _syn.write_fastq(f1, hs1[: n * L], n, L, 1, start=rep * n, append=rep > 0); _syn.write_fastq(f2, hs2[: n * L], n, L, 2, start=rep * n, append=rep > 0)
        thr = max(1, min(args.e2e_threads, cores))
        batch = 1 << 18
        runs = {}
        for names in (True, False):
            t = time.perf_counter()
            st = ra.MappedStream(qi, f1, f2, opts=opts, device=dev_id, batch_units=batch, threads=thr, ph_compact=args.ph_compact, names=names)
            nh = 0
            for b_ in st:
                nh += b_.n_hits
            dt_e = time.perf_counter() - t
            ss = st.stats(); st.close()
            runs[names] = (dt_e, ss, nh)
        dt_e, ss, nh = runs[True]
        out["end_to_end"] = {
            "value": round(ne / dt_e / 1e6, 3), "unit": "M read-pairs/s", "pairs": ne, "hits": int(nh),
            "input": "two plain FASTQ files, %d MB together, in %s (%s), read once right after they were written" % (
                (os.path.getsize(f1) + os.path.getsize(f2)) >> 20, d_e, fstype_of(d_e)),
            "ingest_threads": thr, "batch_units": batch, "names_kept": True,
            "pinned_pool": "1280 MB reserved with qm_stream_reserve before the timed region (pinned in the background, as the CLI does under the index upload)",
            "seconds": {"total_open_to_last_batch_handed_out": round(dt_e, 4), "open": round(ss["open_s"], 4),
                        "first_batch_packed": round(ss["first_batch_s"], 4), "ingest_open_to_last_batch_packed": round(ss["read_s"], 4),
                        "last_batch_mapped": round(ss["last_mapped_s"], 4),
                        "upload_and_kernels_summed_over_contexts": round(ss["map_s"], 4), "download_summed_over_contexts": round(ss["fetch_s"], 4),
                        "caller_waiting": round(ss["caller_wait_s"], 4), "parse_tasks_cpu": round(ss["parse_cpu_s"], 4),
                        "copy_tasks_cpu": round(ss["copy_cpu_s"], 4)},
            "without_read_names": {"value": round(ne / runs[False][0] / 1e6, 3), "seconds_total": round(runs[False][0], 4)},
            "what": "qm_stream_*: ingest workers parse the files chunk-parallel and pack batches straight into pinned slots, device contexts "
                    "sharing the index replica upload / map / download, hits handed out in pinned memory in input order; stream open to last batch"}
        if getattr(args, "gz_leg", 0):
            # (3) the same through ORDINARY gzip files (one deflate stream each, `gzip -6`: what real callers have): the
            # first gz_leg pairs of the batch, inflated by several threads per file (rapmap_amd/csrc/qm_pgz.h), next to the single zlib
            # stream the reference reads them with (QM_INGEST_NO_PGZ=1).  Opt-in: gzip takes a minute to WRITE such files.
            import subprocess
            ng = min(n, int(args.gz_leg))
            g1 = os.path.join(d_e, "g1.fq"); g2 = os.path.join(d_e, "g2.fq")
            _syn.write_fastq(g1, hs1[: ng * L], ng, L, 1); _syn.write_fastq(g2, hs2[: ng * L], ng, L, 2)
            ps = [subprocess.Popen("gzip -6 -f %s" % x, shell=True) for x in (g1, g2)]
            assert all(q.wait() == 0 for q in ps)
            leg = {}
            for kind, env in (("several_threads_per_file", {}), ("one_zlib_stream_per_file", {"QM_INGEST_NO_PGZ": "1"})):
                os.environ.update(env)
                best = None
                for _ in range(2):
                    t = time.perf_counter()
                    st = ra.MappedStream(qi, g1 + ".gz", g2 + ".gz", opts=opts, device=dev_id, batch_units=batch, threads=max(thr, min(48, cores)), ph_compact=args.ph_compact, names=False)
                    nh_g = sum(b_.n_hits for b_ in st)
                    dtg = time.perf_counter() - t
                    st.close()
                    best = dtg if best is None or dtg < best else best
                for k_ in env:
                    del os.environ[k_]
                leg[kind] = {"value": round(ng / best / 1e6, 3), "seconds": round(best, 4), "hits": int(nh_g)}
            leg.update(unit="M read-pairs/s", pairs=ng, input="two `gzip -6` files, %d MB together" % ((os.path.getsize(g1 + ".gz") + os.path.getsize(g2 + ".gz")) >> 20),
                       same_hits=leg["several_threads_per_file"]["hits"] == leg["one_zlib_stream_per_file"]["hits"])
            out["end_to_end_gzip"] = leg
            for x in (g1 + ".gz", g2 + ".gz"):
                os.remove(x)
    finally:
        for f_ in (f1, f2):
            if os.path.exists(f_):
                os.remove(f_)
        try:
            os.rmdir(d_e)
        except OSError:
            pass


def other_configs(oc, args, ra, qd, oracles, oracle, qi, idx_dir, mp, s1, s2, off, ptr, n, L, dev_id, device, k, w_dense, bpp_dense):
    """bounded legs (3 timed steps each) of the configurations the headline does not cover, on the same reads"""
    steps, warm = 3, 1
    # the -p index's oracle loads while the oracle of the -s leg maps its sample (its table -- enumerated from the suffix array and the text, 20 s of
    # numpy -- comes from the background child that built the index: q5.load's enum_cache)
    import threading
    ph_box = {}

    def load_ph():
        try:
            ph_box["idx"] = build_or_reuse_index(args.genes, 42, k, 0, 1, args.cache, True, wait_only=args._bg.ok("ph"))      # (built in the background since the run began)
            marker = os.path.join(os.path.dirname(ph_box["idx"]), "DONE_ORACLE")
            while args._bg.ps.get("ph") is not None and args._bg.ps["ph"].poll() is None and not os.path.exists(marker):
                time.sleep(0.25)                             # (the child is still enumerating the oracle's table: q5.load's enum_cache)
            ph_box["orc"] = oracles.get(ph_box["idx"])
        except Exception as ex:  # noqa: BLE001
            ph_box["err"] = ex
    ph_thread = threading.Thread(target=load_ph, daemon=True)
    ph_thread.start()

    def leg(name, mapper, o, key, bpp, w, cp, workload, extra=None, whole_step=False):
        ph_thread.join()                                     # (no timed step beside a thread of this process that holds the interpreter lock for long stretches)
        with args._bg.quiet():
            el, kms, tot = timed_steps(mapper, o, ptr, n, L, steps, warm, 1, device, qd)
        val = n * steps / el / 1e6
        km = float(np.mean(kms))
        oc[name] = {"workload": workload, "value": round(val, 4), "unit": "M read-pairs/s", "steps": steps, "warmup": warm,
                    "ms_per_step": round(el / steps * 1e3, 3), "kernel_ms": round(km, 3),
                    "hits_per_pair": round(tot["totHits"] / max(1, tot["numReads"]), 4),
                    "roofline": roofline(bpp, w, n, km, KERNELS[key], key, args.genes, step_ms=el / steps * 1e3,
                                         extra=(extra(val, el / steps * 1e3) if callable(extra) else extra), whole_step=whole_step),
                    "parity": {"sample_pairs": cp["sample"], "bit_identical_to_oracle": cp["parity"], "hits": cp["hits"]},
                    "cpu_baseline": {"value": round(cp["cpu_val"], 5), "unit": "M read-pairs/s", "cores": cp["best_t"], "kind": "port"}}
        log("other_configs %s: %.1f M pairs/s, kernel %.2f ms, parity %s" % (name, val, km, cp["parity"]))

    # configs[4]: -s on the dense index (the headline's mapper, selective-alignment options)
    o_sel = ra.default_opts(sel_aln=1)
    oo_sel = oracle.default_opts(selAln=1)
    mp.map_device(min(n, 100000), ptr[0], ptr[1], ptr[2], ptr[3], L, opts=o_sel, fetch=False)      # builds the -s extension table
    # (the oracle maps ~0.45 M pairs/s with -s on this host: 26 s cover the whole batch of 10 M pairs, so that the leg's parity is the
    # full batch like the other legs')
    cp = cpu_and_parity(oracles.get(idx_dir), mp, o_sel, oo_sel, s1, s2, off, ptr, n, L, max(args.cpu_seconds, 26.0) if n >= 5_000_000 else args.cpu_seconds, sweep=False)
    bpp, w = algorithmic_bytes_per_pair(cp["work"], cp["sample"], L)
    leg("configs[4] selective alignment (-s)", mp, o_sel, "sel", bpp, w, cp,
        "the headline's index and reads with -s: chain-scoring collector, chaining, ksw2 extension alignment, score gate",
        extra=lambda val, step_ms: {"dp": dp_roofline(w, val * 1e6, step_ms, n, args.genes, mp),
                                    "note": "kernel_ms spans the two stage-A launches (collector, then intervals -> lists); the plan / ksw2 / finish "
                                            "kernels of stage B-C are in ms_per_step, which is what `frac` is taken over"},
        whole_step=True)

    # configs[3]: the same transcriptome indexed with -p, in both device images, same reads (the text is the same)
    ph_thread.join()
    if "err" in ph_box:
        raise ph_box["err"]
    idx_ph = ph_box["idx"]
    qi_ph = ra.QuasiIndex(idx_ph)
    assert qi_ph.text_len == qi.text_len and qi_ph.n_txps == qi.n_txps
    o = ra.default_opts()
    orc_ph = ph_box["orc"]
    mp_c = ra.QuasiMapper(qi_ph, dev_id, ph_compact=True)
    cp = cpu_and_parity(orc_ph, mp_c, o, oracle.default_opts(), s1, s2, off, ptr, n, L, max(args.cpu_seconds, 30.0), sweep=False)
    bpp, w = algorithmic_bytes_per_pair(cp["work"], cp["sample"], L)
    add, lv = ph_walk_addend(idx_ph, w["n_probe"])
    leg("configs[3] -p index, BooPHF walked on the device (QM_CTX_PH_COMPACT)", mp_c, o, "ph_compact", bpp + add, w, cp,
        "`quasiindex -p` index of the same transcriptome, FrugalBooMap probe path kept on the device", extra={"ph_levels_per_probe": round(lv, 3)})
    mp_c.close()
    mp_e = ra.QuasiMapper(qi_ph, dev_id)
    gr = mp_e.map_device(cp["sample"], ptr[0], ptr[1], ptr[2], ptr[3], L, opts=o, fetch=True)
    ref = cp["ores"]                                          # the oracle's hits on the -p index, computed for the leg above
    cp2 = dict(cp); cp2["parity"] = bool(np.array_equal(gr.hit_offsets, ref.hit_offsets) and gr.hits.tobytes() == ref.hits.tobytes())
    leg("configs[3] -p index, expanded into the bucket table at load (default image)", mp_e, o, "ph_expanded", bpp, w, cp2,
        "the same -p index answered from the one-sector bucket table built from it at load: the dense kernel, the dense algorithmic bytes")
    mp_e.close()
    qi_ph.close()


if __name__ == "__main__":
    main()
