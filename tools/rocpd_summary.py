#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2 rocpd sqlite output) into text: per-kernel time stats from a
--kernel-trace --stats run and per-kernel PMC averages from --pmc runs.

  python tools/rocpd_summary.py trace gpurun_out/prof_trace/bench_results.db
  python tools/rocpd_summary.py pmc   gpurun_out/prof_pmc_fetch/bench_results.db [...]
  python tools/rocpd_summary.py json  [--build-id=ID] c2:N200:B256 rollout_kernel <pmc dbs...>   (merge into profiles/pmc_*.json;
                                       "pair_tile_kernel*30,point_pass_kernel*30": a launch made of several kernels)
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


def dispatches(path, substr):
    """Every dispatch of the kernels whose name contains `substr`, in launch order: duration and grid."""
    cur = sqlite3.connect(path).cursor()
    cols = [d[0] for d in cur.execute("select * from kernels limit 1").description]
    name = "kernel_name" if "kernel_name" in cols else "name"
    grid = [c for c in ("grid_x", "grid_size_x", "grid_size") if c in cols]
    q = f"select {name}, start, end" + (f", {grid[0]}" if grid else "") + " from kernels order by start"
    t0 = None
    print(f"# dispatches matching '{substr}'  ({path})")
    for row in cur.execute(q):
        if t0 is None:
            t0 = row[1]
        if substr in row[0]:
            g = f" grid {row[3]}" if grid else ""
            print(f"{(row[1] - t0) / 1e3:10.1f} us  +{(row[2] - row[1]) / 1e3:9.1f} us{g}  {row[0][:60]}")


def _has(cur, table, col):
    try:
        cols = [d[0] for d in cur.execute(f"select * from {table} limit 1").description]
        return col in cols
    except Exception:
        return False


def pmc(paths):
    import os
    for path in paths:
        if not os.path.exists(path):
            print(f"# (no output: {path} -- that counter pass failed)")
            continue
        cur = sqlite3.connect(path).cursor()
        acc = defaultdict(lambda: defaultdict(float))
        try:
            rows = cur.execute("select kernel_name, dispatch_id, counter_name, value from counters_collection").fetchall()
        except sqlite3.Error as e:
            print(f"# (unreadable: {path}: {e})")
            continue
        for kname, disp, cname, val in rows:
            acc[(kname, cname)][disp] += val            # sum over dimensions (XCC / SE / instances)
        print(f"# PMC per-dispatch averages (summed over hardware instances)  ({path})")
        print(f"{'kernel':70s} {'counter':22s} {'dispatches':>10s} {'avg_per_dispatch':>18s} {'max':>16s}")
        for (kname, cname), d in sorted(acc.items()):
            vals = list(d.values())
            print(f"{kname[:70]:70s} {cname:22s} {len(vals):10d} {sum(vals)/len(vals):18.2f} {max(vals):16.2f}")


def merge_json(key, kernel_spec, paths, build_id=None):
    """Per-launch counter values -> profiles/pmc_counters.json[key] (all counters) and profiles/pmc_traffic.json[key]
    (bytes at the L2 -> fabric boundary per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB; the factor 2 on gfx950 is
    calibrated on known byte counts: profiles/r03_fetch_size_calibration.txt).
    kernel_spec: "substr" = average per dispatch of the kernels whose name contains it (one launch = one dispatch), or
    "substrA*30,substrB*30" = sum over the listed kernels of (average per dispatch x count): a launch made of several
    kernels per horizon step (the batch-major path).  build_id (gpmpc_build_id of the profiled library) is stored beside
    the values: bench.py nulls counter-derived figures when the loaded library is another build."""
    import json
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    parts = []
    for item in kernel_spec.split(","):
        sub, _, mult = item.partition("*")
        parts.append((sub, float(mult) if mult else 1.0))
    vals = {}
    for path in paths:
        cur = sqlite3.connect(path).cursor()
        acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
        for kname, disp, cname, val in cur.execute(
                "select kernel_name, dispatch_id, counter_name, value from counters_collection"):
            for sub, _ in parts:
                if sub in kname:
                    acc[sub][cname][disp] += val
                    break
        for sub, mult in parts:
            for cname, d in acc[sub].items():
                # the largest launches only: warm-up / timing launches of other batch sizes share the kernel name
                v = sorted(d.values())
                top = [x for x in v if x >= 0.5 * v[-1]] if v[-1] > 0 else v
                vals[cname] = vals.get(cname, 0.0) + mult * sum(top) / len(top)
    if not vals:
        raise SystemExit(f"no dispatch of a kernel matching {kernel_spec!r}")
    vals["_kernels"] = kernel_spec
    if build_id:
        vals["_build_id"] = build_id

    def update(fname, value):
        fp = os.path.join(root, fname)
        try:
            cur = json.load(open(fp))
        except Exception:
            cur = {}
        cur[key] = value
        json.dump(cur, open(fp, "w"), indent=1, sort_keys=True)
    update("pmc_counters.json", vals)
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        update("pmc_traffic.json", {"bytes": int((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024), "build_id": build_id,
                                    "kernels": kernel_spec})
    print(key, vals)


if __name__ == "__main__":
    if sys.argv[1] == "trace":
        trace(sys.argv[2])
    elif sys.argv[1] == "list":
        dispatches(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "json":
        args = sys.argv[2:]
        bid = None
        if args[0].startswith("--build-id="):
            bid = args.pop(0).split("=", 1)[1]
        merge_json(args[0], args[1], args[2:], build_id=bid)
    else:
        pmc(sys.argv[2:])
