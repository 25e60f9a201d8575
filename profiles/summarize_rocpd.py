#!/usr/bin/env python3
"""Turn a rocprofv3 (rocpd sqlite) result into the text summary committed under profiles/.
usage: summarize_rocpd.py trace_results.db > profiles/rNN_kernel_stats.txt"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
print("# rocprofv3 --kernel-trace --stats summary (durations in ms; source: %s)" % sys.argv[1])
print("%-100s %8s %14s %12s %8s" % ("kernel", "calls", "total_ms", "avg_ms", "pct"))
for name, calls, tot, avg, pct in c.execute(
        "select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc limit 15"):
    # the view reports microseconds
    print("%-100s %8d %14.3f %12.3f %8.2f" % (name[:100], calls, tot / 1e3, avg / 1e3, pct))
print()
print("# per-dispatch rows of the mapping kernel (start/end are ns timestamps)")
for r in c.execute("select name, grid_x, workgroup_x, lds_size, vgpr_count, sgpr_count, scratch_size, (end-start)/1e6 "
                   "from kernels where name like '%qm_read_kernel%' order by start"):
    print("%s grid=%d wg=%d lds=%d vgpr=%d sgpr=%d scratch=%d dur_ms=%.3f" % r)
