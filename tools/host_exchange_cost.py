#!/usr/bin/env python3
"""What `bench.py --exchange host` costs the HOST per step at world size W: W gloo processes on this
box's CPU cores, each running the real sharding.PendingBest.result() on a (2 + H A)-double record -- clone + one gloo
all_gather_into_tensor of W records + the cross-rank keep-the-best rule -- `iters` times, barrier-free (each rank enters
the next exchange as soon as it has the previous result, like bench.py's loop).  Prints one JSON object (rank 0):
median / p90 / p99 / max of the per-call host time in ms, per rank.

  python tools/host_exchange_cost.py --world 8 --iters 1000 --out profiles/r04_host_exchange_world8.json
(gloo prints its connection banners to stdout: the JSON goes to --out)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, iters, HA, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gp_mpc_amd import sharding
    rng = np.random.default_rng(rank)
    H, A = HA, 1
    xg = sharding.host_group(None)
    times = np.empty(iters + 50)
    for k in range(iters + 50):
        rec = torch.zeros(2 + HA, dtype=torch.float64)
        rec[0], rec[1] = float(rng.uniform()), float(rank * 256 + rng.integers(256))
        pend = sharding.PendingBest(rec, None, world, H, A, rec, exchange_group=xg)
        t0 = time.perf_counter()
        pend.result()
        times[k] = time.perf_counter() - t0
    t = times[50:] * 1e3
    ret[rank] = dict(median=float(np.median(t)), p90=float(np.percentile(t, 90)), p99=float(np.percentile(t, 99)), max=float(t.max()),
                     mean=float(t.mean()))
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--iters", type=int, default=1000)
    ap.add_argument("--record", type=int, default=25, help="H*A of the winner record (config 2: 25)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(worker, args=(args.world, 29000 + os.getpid() % 2000, args.iters, args.record, ret), nprocs=args.world, join=True)
    per = [ret[r] for r in range(args.world)]
    out = {"what": "host time of sharding.PendingBest.result() with a gloo exchange group (clone + all_gather_into_tensor of "
                   f"{args.world} x {2 + args.record} doubles + keep-the-best rule), {args.iters} calls per rank after 50 warm-up calls, ms",
           "world": args.world, "record_doubles": 2 + args.record, "cpu_count": os.cpu_count(),
           "load_note": "the ranks share this box's cores with nothing else; on a GPU node each rank also drives its GPU",
           "median_ms_worst_rank": max(p["median"] for p in per), "p99_ms_worst_rank": max(p["p99"] for p in per),
           "per_rank": per}
    if args.out:
        json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
