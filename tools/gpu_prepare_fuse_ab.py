"""A/B of the 32-wide panel path of gpmpc_prepare (240 < N < 640): trailing update fused with the next diagonal block's factorisation
(option prepare_fuse = 1) against separate launches (0) -- time of a full factorisation, accuracy against the CPU oracle, and the
factors of the two forms compared bit for bit.  Run with GPMPC_LIB pointing at the library under test."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import gp_mpc_amd
from oracle import synth
from oracle import gpmpc_oracle as orc
from helpers import rel_err
shapes = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(257, 2, 1), (300, 3, 1), (333, 3, 2), (400, 4, 2), (500, 2, 1), (500, 4, 2), (544, 3, 1), (600, 4, 2)]
eng = gp_mpc_amd.HipEngine(0)
print("library", gp_mpc_amd.LIB_PATH, "build", eng.build_id)
eng.set_option("incremental", 0)
for (N, D, A) in shapes:
    w = synth.make_workload(N, D, A, 2, 2, seed=1)
    X, Y = torch.as_tensor(w.X).cuda(), torch.as_tensor(w.Y).cuda()
    ls, osc, nz = torch.as_tensor(w.lengthscales).cuda(), torch.as_tensor(w.outputscales).cuda(), torch.as_tensor(w.noises).cuda()
    iK0, beta0 = orc.factorize(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    got = {}
    # (fuse, inv): inv = 2 one-launch inverse, 4 / 1 row blocks per side-stream launch with the default crossover switched off
    for fuse, inv in ((1, 2), (1, 4), (1, 1), (0, 1), (1, 2), (1, 4), (1, 1), (0, 1)):
        eng.set_option("prepare_fuse", fuse)
        eng.set_option("prepare_invcols", 2 if inv == 2 else 0)
        eng.set_option("prepare_inv_batch", 4 if inv == 4 else 1)
        eng.prepare(X, Y, ls, osc, nz); torch.cuda.synchronize()
        ts = []
        for _ in range(20):
            t0 = time.perf_counter(); eng.prepare(X, Y, ls, osc, nz); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        iK, beta = eng.factors()
        got[(fuse, inv)] = (iK.clone(), beta.clone())
        print(f"N={N} D={D} prepare_fuse={fuse} inverse={ {2: 'one launch', 4: '4 row blocks per launch', 1: 'a launch per row block'}[inv] }: {np.median(ts)*1e3:.3f} ms (min {min(ts)*1e3:.3f})  rel err iK {rel_err(iK.cpu().numpy(), iK0):.1e} "
              f"beta {rel_err(beta.cpu().numpy(), beta0):.1e}", flush=True)
    same = lambda p, q: bool(torch.equal(got[p][0], got[q][0]) and torch.equal(got[p][1], got[q][1]))
    print(f"N={N} D={D}: factors identical: fused vs separate {same((1, 1), (0, 1))}, one-launch inverse vs row blocks {same((1, 2), (1, 1))}, "
          f"batched vs single row blocks {same((1, 4), (1, 1))}", flush=True)
eng.set_option("prepare_fuse", 1)
eng.set_option("prepare_invcols", 1)
eng.set_option("prepare_inv_batch", 4)
eng.close()
