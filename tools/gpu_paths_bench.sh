#!/bin/bash
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "paths_agree or large_input" > gpurun_out/pytest_paths.log 2>&1; tail -8 gpurun_out/pytest_paths.log
timeout 300 python -u - <<'PY' 2>&1 | tee gpurun_out/paths_bench.log
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import gp_mpc_amd
from oracle import synth
eng = gp_mpc_amd.HipEngine(0)
for name, B in [("c2", 256), ("c1", 256), ("c3", 256), ("c4", 256)]:
    w = synth.named(name, B=B)
    eng.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    eng.set_cost(w.target, w.W, w.W_T, w.kappa)
    for fp in (0, 3, 2, 1):
        eng.set_option("force_path", fp % 3); eng.set_option("force_separable", int(fp == 3))
        ms, J = eng.rollout_timed(w.actions, w.mu0, w.S0, 2)
        ms, J = eng.rollout_timed(w.actions, w.mu0, w.S0, 5)
        print(f"{name} B={B} force_path={fp}: {ms:.3f} ms/launch -> {B/ms*1e3:.0f} rollouts/s  ({ms*1e3/w.actions.shape[1]:.1f} us/step) J[0]={float(J[0]):.12g}", flush=True)
    eng.set_option("force_path", 0); eng.set_option("force_separable", 0)
PY
