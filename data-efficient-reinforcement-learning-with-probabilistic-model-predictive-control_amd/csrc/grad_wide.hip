// grad_wide.hip -- host side of the analytic gradient for state dimensions beyond 8 (grad_wide_kernel.h).
#include <cstring>

#include "grad_wide_kernel.h"

namespace gpmpc_hip {

// State dimensions beyond 8 (config 5): the matrix-core moment pass + the pair-walking reverse sweep of grad_wide_kernel.h.
int launch_rollout_grad_wide(Handle* h, RolloutArgs& a, double* grad_out, hipStream_t s) {
    const int N = a.N, D = a.D, A = a.A, E = a.E, H = a.H, B = a.B;
    const int NX = E - D, P = D * (D + 1) / 2;
    if (D > 16 || NX > 16) { h->err = "gradient: supported for D <= 16"; return GPMPC_ERR_LIMIT; }
    if (B > 65535 || H > 65535) { h->err = "gradient (D > 8): at most 65535 candidates per launch"; return GPMPC_ERR_LIMIT; }
    const int NSP = wide_nsp(D, NX);
    const WideMomLayout ML = make_wide_mom_layout(N, D, E, NSP);
    const WideSweepLayout SL = make_wide_sweep_layout(D, A, E);
    if ((size_t)ML.total * 8 > (size_t)h->lds_limit) { h->err = "gradient (D > 8): N too large for the per-point factor arrays in LDS"; return GPMPC_ERR_LIMIT; }
    if ((size_t)SL.total * 8 > (size_t)h->lds_limit) { h->err = "gradient (D > 8): sweep LDS layout too large"; return GPMPC_ERR_LIMIT; }
    const size_t n_mom = (size_t)B * H * P * NSP, n_cv = (size_t)B * (H + 1);
    int rc = grow(h, h->gradws, n_mom + n_cv);
    if (rc) return rc;
    if (!a.cv_out) a.cv_out = h->gradws.p + n_mom;
    rc = launch_rollout(h, a, s);            // forward: trajectory, costs, J
    if (rc) return rc;
    WideArgs w;
    memset(&w, 0, sizeof w);
    w.Xt = a.Xt; w.beta = a.beta; w.iK = h->iK.p; w.ils2 = a.ils2; w.var = a.var; w.logvar = a.logvar; w.cost = a.cost;
    w.actions = a.actions; w.mu = a.mu_out; w.Sig = a.Sig_out; w.cv = a.cv_out;
    w.mom = h->gradws.p; w.grad = grad_out; w.kappa = a.kappa; w.use_constraints = a.use_constraints;
    w.N = N; w.D = D; w.A = A; w.E = E; w.H = H; w.B = B; w.include_time = a.include_time; w.time0 = a.time0; w.NSP = NSP;
    w.xrange = h->xrange.p; w.force_path = h->opt_force_path;
    {
        auto kern = wide_pair_moments_kernel<16>;
        rc = allow_full_lds(h, reinterpret_cast<const void*>(kern));
        if (rc) return rc;
        hipLaunchKernelGGL(kern, dim3(P, H, B), dim3(kWideThreads), (size_t)ML.total * 8, s, w);
        GPMPC_HIP_CHECK(h, hipGetLastError());
    }
    {
        auto kern = wide_adjoint_sweep_kernel<16>;
        rc = allow_full_lds(h, reinterpret_cast<const void*>(kern));
        if (rc) return rc;
        hipLaunchKernelGGL(kern, dim3(B), dim3(kWideSweepThreads), (size_t)SL.total * 8, s, w);
        GPMPC_HIP_CHECK(h, hipGetLastError());
    }
    return GPMPC_OK;
}

}  // namespace gpmpc_hip
