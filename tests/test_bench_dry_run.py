"""Tier 3 (CPU): `bench.py --dry-run` under the launch contract of the driver -- `python -m torch.distributed.run --nnodes=1
--nproc-per-node 8 ... bench.py --gpus 8 ...` -- with gloo ranks and the stand-in engine of tests/stub_engine.py.  No multi-GPU
node has been available to any round: this keeps the N > 1 code path of the bench (slice arithmetic at B / G = 1024, the exchange
of the winner records, the barrier / max-over-ranks timing, the per_rank block, ONE JSON line from rank 0) exercised before the
first real 8-GPU run.  The numbers of a dry run mean nothing and the line says so."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world, extra, port):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
           "--dry-run"] + extra
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                  # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def test_strong_scaling_line_of_config_5_at_world_8():
    from gp_mpc_amd import sharding
    d = _run(8, ["--workload", "c5", "--candidates-total", "8192"], 29611 + os.getpid() % 300)
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["steps"] == 3 and d["warmup"] == 1
    assert d["config"]["B_total"] == 8192 and d["config"]["B_per_gpu"] == 1024 and d["config"]["N"] == 4096
    assert d["metric"] == "MPC trajectory rollouts/sec" and d["unit"] == "rollouts/s" and d["data"].startswith("DRY RUN")
    assert abs(d["value"] - 8192 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-9          # whole-job aggregate over the 8 ranks
    pr = d["per_rank"]
    assert [r["rank"] for r in pr] == list(range(8))
    assert [r["candidates"] for r in pr] == [sharding.shard_bounds(8192, 8, r)[1] - sharding.shard_bounds(8192, 8, r)[0] for r in range(8)]
    assert sum(r["candidates"] for r in pr) == 8192 and 0 <= d["best_index"] < 8192
    for k in ("roofline", "windows", "closed_loop_ms_per_step", "prepare_ms"):
        assert k in d


def test_weak_scaling_line_and_ragged_strong_split_at_world_3():
    d = _run(3, ["--workload", "c2"], 29911 + os.getpid() % 300)
    assert d["scaling"] == "weak" and d["config"]["B_total"] == 3 * 256 and d["config"]["B_per_gpu"] == 256
    d = _run(3, ["--workload", "c4", "--candidates-total", "1000"], 30211 + os.getpid() % 300)
    assert d["scaling"] == "strong" and [r["candidates"] for r in d["per_rank"]] == [334, 333, 333]
    assert abs(d["value"] - 1000 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-9
