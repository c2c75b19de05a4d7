"""Few-candidate cooperative form of the fused-horizon kernel (option "cluster"): trajectories bit for bit those of the
one-workgroup-per-candidate kernel, and the time per launch, over shapes x candidates x cluster sizes.
  python tools/gpu_cluster_check.py [quick]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import gp_mpc_amd
from oracle import synth

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
eng = gp_mpc_amd.HipEngine(0)


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


shapes = [("c2", 200, 3, 1, 25, False), ("c1", 50, 3, 1, 15, False), ("c3", 500, 2, 1, 40, False), ("d4", 300, 4, 2, 12, True),
          ("n130d3", 130, 3, 2, 10, False), ("d1", 90, 1, 1, 8, False)]
if quick:
    shapes = shapes[:2]
bad = 0
for name, n, d, a, h, tm in shapes:
    w = synth.make_workload(n, d, a, h, 16, include_time=tm, seed=0)
    eng.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    eng.set_cost(w.target, w.W, w.W_T, w.kappa)
    acts = torch.as_tensor(w.actions, device="cuda:0")
    for B in ((1, 2, 9, 16) if not quick else (1, 9)):
        for rpc in ((8, 16, 32) if not quick else (8,)):
            eng.set_option("rows_per_chunk", rpc)
            eng.set_option("cluster", 1)
            ref = eng.rollout(acts[:B], w.mu0, w.S0, w.include_time, w.time0)
            t_ref = timed(lambda: eng.rollout(acts[:B], w.mu0, w.S0, w.include_time, w.time0))
            assert eng.last_cluster == 1
            line = f"{name} N={n} D={d} H={h} B={B} rows/chunk {rpc}: plain {t_ref:.3f} ms |"
            for cs in (0, 2, 4, 6, 8, 12, 16, 32):
                eng.set_option("cluster", cs)
                out = eng.rollout(acts[:B], w.mu0, w.S0, w.include_time, w.time0)
                used = eng.last_cluster
                same = all(torch.equal(out[k], ref[k]) for k in ("mu", "Sig", "J"))
                if not same:
                    bad += 1
                    err = float((out["Sig"] - ref["Sig"]).abs().max())
                    line += f" cs={cs}->{used} MISMATCH (max |dSig| {err:.2e})"
                    continue
                t = timed(lambda: eng.rollout(acts[:B], w.mu0, w.S0, w.include_time, w.time0))
                line += f" cs={cs}->{used} {t:.3f}"
            print(line, flush=True)
        # the defaults of both forms (their own chunk lengths): equal to the method's noise floor, not bit for bit
        eng.set_option("rows_per_chunk", 0)
        eng.set_option("cluster", 1)
        ref = eng.rollout(acts[:B], w.mu0, w.S0, w.include_time, w.time0)
        t_ref = timed(lambda: eng.rollout(acts[:B], w.mu0, w.S0, w.include_time, w.time0))
        eng.set_option("cluster", 0)
        out = eng.rollout(acts[:B], w.mu0, w.S0, w.include_time, w.time0)
        t = timed(lambda: eng.rollout(acts[:B], w.mu0, w.S0, w.include_time, w.time0))
        e_mu = float((out["mu"] - ref["mu"]).abs().max() / ref["mu"].abs().max())
        e_S = float((out["Sig"] - ref["Sig"]).abs().max() / ref["Sig"].abs().max())
        print(f"{name} B={B} defaults: plain {t_ref:.3f} ms, auto -> cluster {eng.last_cluster} {t:.3f} ms; rel diff mu {e_mu:.1e} Sig {e_S:.1e}", flush=True)
        if e_mu > 1e-9 or e_S > 1e-5:
            bad += 1
    # many launches back to back (tags, buffer reuse), each compared
    eng.set_option("cluster", 0)
    eng.set_option("rows_per_chunk", 8)
    for i in range(100 if not quick else 20):
        B = 1 + i % 3
        out = eng.rollout(acts[:B], w.mu0, w.S0, w.include_time, w.time0)
        eng.set_option("cluster", 1)
        ref = eng.rollout(acts[:B], w.mu0, w.S0, w.include_time, w.time0)
        eng.set_option("cluster", 0)
        if not (torch.equal(out["mu"], ref["mu"]) and torch.equal(out["Sig"], ref["Sig"])):
            bad += 1
            print(f"{name}: repeat {i} B={B} MISMATCH", flush=True)
            break
    eng.set_option("rows_per_chunk", 0)
    # objective + gradient at B = 1
    for cs in (1, 0):
        eng.set_option("cluster", cs)
        g = eng.rollout_grad(acts[:1], w.mu0, w.S0, w.include_time, w.time0)
        t = timed(lambda: eng.rollout_grad(acts[:1], w.mu0, w.S0, w.include_time, w.time0))
        print(f"{name}: rollout_grad B=1 cluster option {cs} -> {eng.last_cluster}: {t:.3f} ms, J {float(g['J'][0]):.12g}", flush=True)
    eng.set_option("cluster", 0)
print("MISMATCHES" if bad else "all bitwise equal", bad)
eng.close()
sys.exit(1 if bad else 0)
