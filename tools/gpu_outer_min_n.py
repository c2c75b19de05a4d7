import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch, gp_mpc_amd
from oracle import synth, gpmpc_oracle as orc
from helpers import rel_err
eng = gp_mpc_amd.HipEngine(0); eng.set_option("incremental", 0)
for (N, D) in [(400, 3), (500, 2), (640, 3), (768, 4), (900, 2), (1000, 4)]:
    w = synth.make_workload(N, D, 1, 2, 2, seed=1)
    X, Y = torch.as_tensor(w.X).cuda(), torch.as_tensor(w.Y).cuda()
    ls, osc, nz = torch.as_tensor(w.lengthscales).cuda(), torch.as_tensor(w.outputscales).cuda(), torch.as_tensor(w.noises).cuda()
    out = []
    for mn in (1024, 256):
        eng.set_option("outer_min_n", mn)
        eng.prepare(X, Y, ls, osc, nz); torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); eng.prepare(X, Y, ls, osc, nz); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        iK, beta = eng.factors()
        iK0, beta0 = orc.factorize(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
        out.append(f"min_n={mn}: {np.median(ts)*1e3:.3f} ms (err {rel_err(iK.cpu().numpy(), iK0):.1e})")
    print(N, D, " | ".join(out), flush=True)
