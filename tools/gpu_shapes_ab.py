"""A/B of two builds of the library on the BASELINE shapes (subprocess per build, clocks pre-conditioned)."""
import os, subprocess, sys
libs = sys.argv[1:]
code = r'''
import sys, time
sys.path.insert(0, '.')
import gp_mpc_amd
from oracle import synth
eng = gp_mpc_amd.HipEngine(0)
for name, B in [("c2", 256), ("c4", 128)]:
    w = synth.named(name, B=B)
    eng.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    eng.set_cost(w.target, w.W, w.W_T, w.kappa)
    t0 = time.time()
    while time.time() - t0 < 0.3:
        eng.rollout_timed(w.actions, w.mu0, w.S0, 2, w.include_time, w.time0)
    best = min(eng.rollout_timed(w.actions, w.mu0, w.S0, 5, w.include_time, w.time0)[0] for _ in range(4))
    print(f"  {name} B={B}: {best:.3f} ms/launch", flush=True)
'''
for rnd in range(1):
    for lib in libs:
        print(lib, flush=True)
        env = dict(os.environ, GPMPC_LIB=os.path.abspath(lib))
        subprocess.run([sys.executable, "-c", code], env=env, stderr=subprocess.DEVNULL)
