#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the passes of profiles/r06/pmc_all.sh.
usage (on the GPU box, behind pmc_all.sh): python profiles/r06/make_pmc_traffic.py gpurun_out/<dir> <tag> [outdir]
writes pmc_traffic.json and the three summaries (named as they are committed under profiles/r06/) into outdir (default: the
repository's profiles/ and profiles/r05/); with an outdir under gpurun_out/ they travel back and are copied into place by hand"""
import collections, csv, glob, json, os, shutil, sys

src, tag = sys.argv[1], sys.argv[2]
outdir = sys.argv[3] if len(sys.argv) > 3 else None
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N = 10_000_000


def counters(d):
    tot = collections.defaultdict(collections.Counter); cnt = collections.defaultdict(collections.Counter)
    for f in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "qm" not in k: continue
            k = k.split("(")[0].replace("void ", "")
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    return {k: {c: tot[k][c] / cnt[k][c] for c in tot[k]} for k in tot}, {k: max(cnt[k].values()) for k in cnt}


def trace_ms(d):
    """kernel name -> (calls, average ms) from the --stats run (3 timed steps + 1 warm-up)"""
    out = {}
    for f in glob.glob(os.path.join(src, d, "stats", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "qm" in r["Name"]:
                out[r["Name"].split("(")[0].replace("void ", "")] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6)
    return out


def hbm(c):
    return c.get("FETCH_SIZE", 0) * 1024 + c.get("WRITE_SIZE", 0) * 1024


out = {}
names = {"dense_pair": "kernel_stats_and_pmc_dense_pair_kernel_%s.txt", "dense_lean": "kernel_stats_and_pmc_dense_lean_kernel_%s.txt", "ph_compact": "kernel_stats_and_pmc_ph_compact_%s.txt",
         "sel": "kernel_stats_and_pmc_sel_%s.txt", "default_two_parts": "kernel_stats_default_two_parts_%s.txt"}
for key in names:
    shutil.copy(os.path.join(src, key, "summary.txt"), os.path.join(outdir or os.path.join(ROOT, "profiles", "r06"), names[key] % tag))


def entry(key, match):
    C, _ = counters(key); T = trace_ms(key)
    k = [x for x in C if match in x][0]
    c = C[k]
    f = "profiles/r06/" + names[key] % tag
    return {"hbm_bytes_per_launch": hbm(c), "sectors_per_pair": c["TCC_MISS_sum"] / N, "kernel": k, "version": "r06 (%s)" % tag,
            "pairs_per_launch": N, "kernel_ms_trace": T[k][1],
            "per_pair": {"SALU": c["SQ_INSTS_SALU"] / N, "VALU": c["SQ_INSTS_VALU"] / N, "BRANCH": c["SQ_INSTS_BRANCH"] / N, "SMEM": c["SQ_INSTS_SMEM"] / N,
                         "LDS": c["SQ_INSTS_LDS"] / N, "VMEM_RD": c["SQ_INSTS_VMEM_RD"] / N, "VMEM_WR": c["SQ_INSTS_VMEM_WR"] / N},
            "note": "FETCH_SIZE + WRITE_SIZE (KiB, separate --pmc passes) of %s, per dispatch, one launch = 10 M pairs (%.2f + %.2f GB; TCC_MISS x 64 B = %.2f GB). "
                    "FETCH_SIZE counts 64 B per TCC_EA read request for this random-sector pattern (the guide's x2 correction is for wide coalesced "
                    "streaming reads, which this kernel does not make)." % (k, c["FETCH_SIZE"] * 1024 / 1e9, c["WRITE_SIZE"] * 1024 / 1e9, c["TCC_MISS_sum"] * 64 / 1e9),
            "source": "profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE of this kernel on this workload (r06, %s); not re-measured by the run that prints it" % f}


pair = entry("dense_pair", "qm_duo_kernel"); lean = entry("dense_lean", "qm_lean_kernel")
# the default step: half of the batch on each kernel
out["dense"] = {"hbm_bytes_per_launch": 0.5 * (pair["hbm_bytes_per_launch"] + lean["hbm_bytes_per_launch"]), "sectors_per_pair": 0.5 * (pair["sectors_per_pair"] + lean["sectors_per_pair"]),
                "kernel": "half of the batch on %s, half on %s, in flight together" % (pair["kernel"], lean["kernel"]), "version": "r06 (%s)" % tag, "pairs_per_launch": N,
                "pair_kernel": pair, "lean_kernel": lean,
                "note": "the default step maps 5 M pairs on each kernel at the same time; the counters were taken with the whole batch of 10 M pairs as ONE launch of either (QM_SPLIT=1; "
                        "QM_NO_DUO=1): hbm_bytes_per_launch / sectors_per_pair here are the means of the two, i.e. the step's",
                "source": "profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE of the two kernels on this workload (r06, profiles/r06/%s and %s); not re-measured by the run that prints it"
                          % (names["dense_pair"] % tag, names["dense_lean"] % tag)}
out["ph_compact"] = entry("ph_compact", "qm_lean_kernel")
out["ph_expanded"] = dict(out["dense"]); out["ph_expanded"]["note"] = "the default image of a -p index is the canonical bucket table: the dense kernel's figure"
C, ND = counters("sel"); T = trace_ms("sel")
f = "profiles/r06/" + names["sel"] % tag
stageA = [k for k in C if "qm_lean_kernel" in k or "h2m" in k or "qm_read_kernel" in k]
# per step: every kernel's per-dispatch counters x its dispatches per step (the PMC runs are one timed step + one warm-up: dispatches / 2)
step_b = 0.0; step_s = 0.0; parts = []
for k in sorted(C, key=lambda k: -hbm(C[k]) * ND[k]):
    if any(x in k for x in ("build_", "unpack")): continue
    per_step = ND[k] / 2.0
    step_b += hbm(C[k]) * per_step; step_s += C[k].get("TCC_MISS_sum", 0) * per_step
    parts.append("%s %.0f x (%.2f + %.2f) GB" % (k.split("::")[-1], per_step, C[k].get("FETCH_SIZE", 0) * 1024 / 1e9, C[k].get("WRITE_SIZE", 0) * 1024 / 1e9))
ksw = sum(T[k][1] * T[k][0] / 4.0 for k in T if "sel_align" in k)      # the --stats run times 4 steps
out["sel"] = {"hbm_bytes_per_launch": sum(hbm(C[k]) for k in stageA), "step_hbm_bytes_per_launch": step_b,
              "sectors_per_pair": sum(C[k].get("TCC_MISS_sum", 0) for k in stageA) / N, "step_sectors_per_pair": step_s / N,
              "ksw2_kernel_ms_per_step": ksw, "kernel": " + ".join(sorted(stageA)), "version": "r06 (%s)" % tag, "pairs_per_launch": N,
              "kernel_ms_trace": {k: {"calls_in_4_steps": T[k][0], "avg_ms": T[k][1]} for k in T if "build_" not in k},
              "note": "FETCH_SIZE + WRITE_SIZE (KiB, separate --pmc passes) per dispatch x dispatches per step of every kernel of an unsplit -s step of 10 M pairs: " + "; ".join(parts),
              "source": "profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE of these kernels on this workload (r06, %s); not re-measured by the run that prints it" % f}
json.dump(out, open(os.path.join(outdir or os.path.join(ROOT, "profiles"), "pmc_traffic.json"), "w"), indent=1)
for k, v in out.items():
    print(k, "%.2f GB" % (v["hbm_bytes_per_launch"] / 1e9), "sectors/pair %.1f" % v["sectors_per_pair"], v.get("per_pair"), v.get("step_hbm_bytes_per_launch"))
