"""TEST INFRASTRUCTURE ONLY -- CPU statement of the GP training loss and its gradient.

The reference trains each GP with gpytorch's ExactMarginalLogLikelihood under torch LBFGS
(rl_gp_mpc/control_objects/models/gp_model.py:193-306; loss = -mll(output, train_y), :262-275).  gpytorch is not
installed in this image, so this path is PARITY-UNPINNED against a reference run: the function below restates
gpytorch 1.x's published definition -- ExactMarginalLogLikelihood.forward returns
MultivariateNormal(0, K + noise I).log_prob(y) / num_data for a ZeroMean ExactGP with ScaleKernel(RBFKernel(ard)) --
and the HIP kernels are checked against it and against torch autograd of it.  Never imported by the product.
"""
import numpy as np


def neg_mll_and_grad(X, y, lengthscale, outputscale, noise):
    """One GP: loss = -log p(y | X, theta) / N and d loss / d (lengthscale (E,), outputscale, noise), closed form."""
    from scipy.linalg import cho_factor, cho_solve
    X = np.asarray(X, float)
    y = np.asarray(y, float)
    N, E = X.shape
    ls = np.asarray(lengthscale, float).reshape(E)
    diff = X[:, None, :] - X[None, :, :]
    d2 = (diff / ls) ** 2                                      # (N,N,E)
    Kp = outputscale * np.exp(-0.5 * d2.sum(-1))
    K = Kp + noise * np.eye(N)
    c = cho_factor(K, lower=True)
    beta = cho_solve(c, y)
    iK = cho_solve(c, np.eye(N))
    logdet = 2.0 * np.log(np.diag(c[0])).sum()
    loss = (0.5 * y @ beta + 0.5 * logdet + 0.5 * N * np.log(2.0 * np.pi)) / N
    Q = np.outer(beta, beta) - iK                              # d mll / dK = Q / 2
    g_ls = -0.5 / N * np.einsum('ij,ij,ije->e', Q, Kp, d2) / ls
    g_os = -0.5 / N * np.sum(Q * Kp) / outputscale
    g_nz = -0.5 / N * np.trace(Q)
    return loss, g_ls, g_os, g_nz


def neg_mll_torch(X, y, lengthscale, outputscale, noise):
    """The same loss as a differentiable torch expression (checker for the closed-form gradient)."""
    import torch
    N = X.shape[0]
    d = (X[:, None, :] - X[None, :, :]) / lengthscale
    K = outputscale * torch.exp(-0.5 * (d * d).sum(-1)) + noise * torch.eye(N, dtype=X.dtype)
    L = torch.linalg.cholesky(K)
    alpha = torch.cholesky_solve(y[:, None], L)[:, 0]
    ll = -0.5 * (y @ alpha) - torch.log(torch.diagonal(L)).sum() - 0.5 * N * np.log(2 * np.pi)
    return -ll / N


def train_loop(X, Y, parameters, constraints, lr, num_iter, seed):
    """The reference's hyper-parameter search restated on the CPU, independent of the product's driver
    (rl_gp_mpc/control_objects/models/gp_model.py:193-306): GP after GP --
      previous loss at the incoming parameters (:226-229); a restart drawn uniformly inside the constraint box in the order
      outputscale, lengthscale, noise (:236-252; gpytorch's Interval constraint maps a raw parameter through a sigmoid, so
      the optimisation variable is raw = logit((value - lo) / (hi - lo))); torch LBFGS with strong-Wolfe line search
      (:256-258), `num_iter` steps, the loss of every step compared with the best so far and the parameters AFTER the step
      kept (:277-282); the incoming parameters win if nothing better was found.
    The reference's clip_grad_value_ call sits BEFORE backward() on freshly zeroed gradients (:266-267) and is a no-op.
    parameters: list of dicts {lengthscale (E,), outputscale, noise}; constraints: dict of (D, ...) arrays
    min/max_lengthscale, min/max_outputscale, min/max_std_noise.  Returns the list of dicts and the list of best losses."""
    import torch
    F = torch.float64
    X = torch.as_tensor(np.asarray(X), dtype=F)
    Y = torch.as_tensor(np.asarray(Y), dtype=F)
    torch.manual_seed(seed)
    out, losses = [], []
    for a, p in enumerate(parameters):
        lo = {"ls": torch.as_tensor(np.asarray(constraints["min_lengthscale"])[a], dtype=F),
              "os": torch.as_tensor(float(np.asarray(constraints["min_outputscale"])[a]), dtype=F),
              "nz": torch.as_tensor(float(np.asarray(constraints["min_std_noise"])[a]) ** 2, dtype=F)}
        hi = {"ls": torch.as_tensor(np.asarray(constraints["max_lengthscale"])[a], dtype=F),
              "os": torch.as_tensor(float(np.asarray(constraints["max_outputscale"])[a]), dtype=F),
              "nz": torch.as_tensor(float(np.asarray(constraints["max_std_noise"])[a]) ** 2, dtype=F)}
        y = Y[:, a]
        best = {"ls": torch.as_tensor(np.asarray(p["lengthscale"], dtype=float).reshape(-1), dtype=F),
                "os": torch.as_tensor(float(p["outputscale"]), dtype=F), "nz": torch.as_tensor(float(p["noise"]), dtype=F)}
        try:
            best_loss = float(neg_mll_torch(X, y, best["ls"], best["os"], best["nz"]))
        except Exception:
            best_loss = float("inf")
        u = {"os": torch.rand((), dtype=F), "ls": torch.rand(best["ls"].shape, dtype=F), "nz": torch.rand((), dtype=F)}
        raw = {k: torch.logit(u[k].clamp(1e-6, 1 - 1e-6)).requires_grad_(True) for k in ("ls", "os", "nz")}

        def value(k):
            return lo[k] + (hi[k] - lo[k]) * torch.sigmoid(raw[k])
        opt = torch.optim.LBFGS([raw["ls"], raw["os"], raw["nz"]], lr=lr, line_search_fn="strong_wolfe")
        try:
            for _ in range(num_iter):
                def closure():
                    opt.zero_grad()
                    loss = neg_mll_torch(X, y, value("ls"), value("os"), value("nz"))
                    loss.backward()
                    return loss
                loss = float(opt.step(closure).detach())
                if loss < best_loss:
                    best_loss = loss
                    best = {k: value(k).detach().clone() for k in raw}
        except Exception:
            pass
        out.append({"lengthscale": best["ls"].numpy().copy(), "outputscale": float(best["os"]), "noise": float(best["nz"])})
        losses.append(best_loss)
    return out, losses
