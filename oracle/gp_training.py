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
