import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gp_mpc_amd
from oracle import synth
N, D, A = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
eng = gp_mpc_amd.HipEngine(0)
for kv in sys.argv[5:]:                      # engine options, name=value
    k, v = kv.split("=")
    eng.set_option(k, float(v))
w = synth.make_workload(N, D, A, 2, 2, seed=1)
X, Y = torch.as_tensor(w.X).cuda(), torch.as_tensor(w.Y).cuda()
ls, osc, nz = torch.as_tensor(w.lengthscales).cuda(), torch.as_tensor(w.outputscales).cuda(), torch.as_tensor(w.noises).cuda()
for _ in range(reps):
    eng.prepare(X, Y, ls, osc, nz)
torch.cuda.synchronize()
print("done")
