#!/usr/bin/env python3
"""profiles/cpu_port_vs_reference.py -- how fast is the CPU port (oracle/qm_oracle.cpp) next to the reference itself?

Runs in the BUILD container only (it needs the survey stage's probe build of the unmodified reference, /tmp/oracle/build/rapmap,
and the survey's 1/10-scale workload under /tmp/oracle/synth: 19 684 transcripts, 1 M pairs 2x100 bp): the reference with
`quasimap -n -t T` (its own "Elapsed time" around the mapping block, src/RapMapSAMapper.cpp:856, index load excluded) and the
oracle's map_pairs on the same index and the same reads with the same thread count.  Writes profiles/port_over_reference.json,
which bench.py attaches to its cpu_baseline object (BASELINE.md section 3's fallback: the port timed on the GPU box, the
port / reference ratio from here)."""
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

REF = "/tmp/oracle/build/rapmap"
W = "/tmp/oracle/synth"


def read_fastq_fast(path):
    seqs = []
    with open(path, "rb") as f:
        for i, l in enumerate(f):
            if i % 4 == 1:
                seqs.append(l.rstrip(b"\n"))
    return seqs


def main():
    from oracle import oracle, q5
    from rapmap_amd import pack_reads
    oracle.build()
    out = {"workload": "survey 1/10-scale set: %s/idx (19 684 transcripts), 1 M pairs 2x100 bp, 1 %% substitutions" % W, "runs": []}
    r1 = read_fastq_fast(W + "/r1.fq"); r2 = read_fastq_fast(W + "/r2.fq")
    q1, o1 = pack_reads(r1); q2, o2 = pack_reads(r2)
    n = len(o1) - 1
    ix = q5.load(W + "/idx")
    orc = oracle.Oracle(ix)
    for T in (1, 8):
        best_ref = None
        for rep in range(2 if T > 1 else 1):
            p = subprocess.run([REF, "quasimap", "-i", W + "/idx", "-1", W + "/r1.fq", "-2", W + "/r2.fq", "-t", str(T), "-n"],
                               capture_output=True, text=True)
            m = re.search(r"Elapsed time: ([0-9.]+)s", p.stdout + p.stderr)
            t = float(m.group(1))
            best_ref = t if best_ref is None else min(best_ref, t)
        best_port = None
        for rep in range(2 if T > 1 else 1):
            t0 = time.perf_counter(); res = orc.map_pairs(q1, o1, q2, o2, nthreads=T); dt = time.perf_counter() - t0
            best_port = dt if best_port is None else min(best_port, dt)
        out["runs"].append({"threads": T, "reference_s": best_ref, "port_s": round(best_port, 3),
                            "reference_Mpairs_s": round(n / best_ref / 1e6, 4), "port_Mpairs_s": round(n / best_port / 1e6, 4),
                            "port_over_reference": round(best_ref / best_port, 3)})
        print(out["runs"][-1], flush=True)
    out["port_over_reference"] = out["runs"][-1]["port_over_reference"]
    out["note"] = ("ratio of throughputs (port / reference) at 8 threads on the build container's 8 cores; the reference is the survey's probe "
                   "build (-O2 -march=native, cereal stand-in, no jemalloc), the port is oracle/libqm_oracle.so (-O2)")
    json.dump(out, open(os.path.join(ROOT, "profiles", "port_over_reference.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
