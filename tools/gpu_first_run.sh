#!/bin/bash
# GPU contact: parity tests + a quick timing of the BASELINE workloads
mkdir -p gpurun_out
timeout 500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
tail -25 gpurun_out/pytest_gpu.log
timeout 200 python -u - <<'PY' 2>&1 | tee gpurun_out/quick_bench.log
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import gp_mpc_amd
from oracle import synth
eng = gp_mpc_amd.HipEngine(0)
for name, B in [("c2", 256), ("c2", 1024), ("c3", 1024), ("c1", 256), ("c4", 256)]:
    w = synth.named(name, B=B)
    eng.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    torch.cuda.synchronize(); t0 = time.time()
    eng.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises); torch.cuda.synchronize(); tp = time.time() - t0
    eng.set_cost(w.target, w.W, w.W_T, w.kappa)
    ms, J = eng.rollout_timed(w.actions, w.mu0, w.S0, 2)
    ms, J = eng.rollout_timed(w.actions, w.mu0, w.S0, 5)
    print(f"{name} B={B}: prepare {tp*1e3:.2f} ms, rollout {ms:.3f} ms/launch -> {B/ms*1e3:.0f} rollouts/s; J[0]={float(J[0]):.10g}", flush=True)
PY
