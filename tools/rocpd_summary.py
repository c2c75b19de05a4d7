#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2 rocpd sqlite output) into text: per-kernel time stats from a
--kernel-trace --stats run and per-kernel PMC averages from --pmc runs.

  python tools/rocpd_summary.py trace gpurun_out/prof_trace/bench_results.db
  python tools/rocpd_summary.py pmc   gpurun_out/prof_pmc_fetch/bench_results.db [...]
"""
import sqlite3
import sys
from collections import defaultdict


def trace(path):
    cur = sqlite3.connect(path).cursor()
    print(f"# kernel-trace stats  ({path})")
    print(f"{'kernel':100s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>7s}")
    for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print(f"{name[:100]:100s} {calls:6d} {total:12.1f} {avg:10.2f} {pct:7.2f}")
    rows = cur.execute("select kernel_name, min(end-start), max(end-start), count(*) from kernels group by kernel_name").fetchall() \
        if _has(cur, "kernels", "kernel_name") else []
    for r in rows:
        print(f"#   {r[0][:80]}: min {r[1]/1e3:.1f} us max {r[2]/1e3:.1f} us n={r[3]}")


def _has(cur, table, col):
    try:
        cols = [d[0] for d in cur.execute(f"select * from {table} limit 1").description]
        return col in cols
    except Exception:
        return False


def pmc(paths):
    for path in paths:
        cur = sqlite3.connect(path).cursor()
        acc = defaultdict(lambda: defaultdict(float))
        for kname, disp, cname, val in cur.execute(
                "select kernel_name, dispatch_id, counter_name, value from counters_collection"):
            acc[(kname, cname)][disp] += val            # sum over dimensions (XCC / SE / instances)
        print(f"# PMC per-dispatch averages (summed over hardware instances)  ({path})")
        print(f"{'kernel':70s} {'counter':22s} {'dispatches':>10s} {'avg_per_dispatch':>18s} {'max':>16s}")
        for (kname, cname), d in sorted(acc.items()):
            vals = list(d.values())
            print(f"{kname[:70]:70s} {cname:22s} {len(vals):10d} {sum(vals)/len(vals):18.2f} {max(vals):16.2f}")


if __name__ == "__main__":
    if sys.argv[1] == "trace":
        trace(sys.argv[2])
    else:
        pmc(sys.argv[2:])
