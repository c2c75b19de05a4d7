"""Time one launch of the streaming rollout kernel at config-5 class shapes (D = 16, A = 4, B = 256 = one workgroup per CU).
  python tools/gpu_c5_step.py [N:H[:s0] ...]      default: 1024:1 2048:1 4096:1 4096:2   (s0: initial state variance, 1e-6);
  name=value arguments are engine options"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gp_mpc_amd
from oracle import synth

shapes = [tuple(float(v) for v in a.split(":")) for a in sys.argv[1:] if "=" not in a] or [(1024, 1), (2048, 1), (4096, 1), (4096, 2)]
eng = gp_mpc_amd.HipEngine(0)
for kv in [a for a in sys.argv[1:] if "=" in a]:        # engine options, name=value (e.g. force_path=4: tabulated exp)
    eng.set_option(kv.split("=")[0], float(kv.split("=")[1]))
for sh in shapes:
    N, H = int(sh[0]), int(sh[1])
    s0 = sh[2] if len(sh) > 2 else 1e-6
    D, A, B = 16, 4, 256
    w = synth.make_workload(N, D, A, H, B, seed=0, s0=s0)
    eng.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    eng.set_cost(w.target, w.W, w.W_T, w.kappa)
    ms, J = eng.rollout_timed(w.actions, w.mu0, w.S0, 1)
    print(f"N={N} D={D} H={H} B={B} s0={s0:g}: {ms:.1f} ms/launch = {ms/H:.1f} ms per horizon step of 256 candidates; J[0]={float(J[0]):.6g}", flush=True)
