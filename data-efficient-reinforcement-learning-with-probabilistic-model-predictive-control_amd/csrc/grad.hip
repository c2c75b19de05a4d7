// grad.hip -- host-side dispatch of the analytic-gradient kernels (grad_kernels.h).
#include "grad_stream_kernel.h"
#include "grad_sep_kernel.h"
#include "moment_schedule.h"
#include <cstring>

namespace gpmpc_hip {

namespace {

constexpr int kMomThreads = 512;

template <int DP, int NXP>
int launch_moments(Handle* h, const GradArgs& g, size_t lds_bytes, hipStream_t s) {
    auto go = [&](auto kern, int nt) -> int {
        int rc = allow_full_lds(h, reinterpret_cast<const void*>(kern));
        if (rc) return rc;
        hipLaunchKernelGGL(kern, dim3(g.H, g.B, g.gz), dim3(nt), lds_bytes, s, g);
        GPMPC_HIP_CHECK(h, hipGetLastError());
        return GPMPC_OK;
    };
    if constexpr (DP <= 3) {
        // two 512-thread workgroups per CU (see pair_moments_kernel), or
        // 16 waves per (candidate, step): the accumulators fit the 128-VGPR budget of a 1024-thread workgroup
        if (g.share_cu) return go(pair_moments_kernel<DP, NXP, 512, 1>, 512);
        return go(pair_moments_kernel<DP, NXP, 1024, 1>, 1024);
    } else {
        return go(pair_moments_kernel<DP, NXP, kMomThreads, 1>, kMomThreads);
    }
}

// Few candidates (B x H x 2 <= CUs: the reference's one-sequence-per-evaluation regime): the element-wise moment pass, the mean
// moments and the stage costs / objective of the stored trajectory are independent of each other and each fills a fraction of
// the chip -- ONE grid runs them side by side (z < gz: pair groups, z = gz: mean moments, z = gz + 1 and x = 0: the candidate's
// costs) instead of three launches in a row (config-2 shape, B = 1: 23 + 7.5 + 12 us and two launch gaps -> ~24 us).
struct CostSlice {
    const double* cost; double kappa; int clip, use_constraints; double* cm; double* cv; double* J;
};
template <int DP, int NXP, int NT>
__global__ __launch_bounds__(NT) void few_candidate_moments_kernel(const GradArgs p, const CostSlice cs) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if ((int)blockIdx.z < p.gz) { pair_moments_body<DP, NXP, NT, 1>(p, smem); return; }
    if ((int)blockIdx.z == p.gz) {
        if ((int)threadIdx.x >= 64 * DP) return;
        mean_moments_body<DP, NXP>(p, smem);
        return;
    }
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;
    traj_cost_body(blockIdx.y, threadIdx.x, p.mu, p.Sig, p.actions, cs.cost, p.D, p.A, p.H, cs.kappa, cs.clip, cs.use_constraints,
                   cs.cm, cs.cv, cs.J);
}

template <int DP, int NXP>
int launch_few_candidate_moments(Handle* h, const GradArgs& g, const CostSlice& cs, size_t lds_bytes, hipStream_t s) {
    constexpr int NT = DP <= 3 ? 1024 : kMomThreads;
    auto kern = few_candidate_moments_kernel<DP, NXP, NT>;
    int rc = allow_full_lds(h, reinterpret_cast<const void*>(kern));
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(g.H, g.B, g.gz + 2), dim3(NT), lds_bytes, s, g, cs);
    GPMPC_HIP_CHECK(h, hipGetLastError());
    return GPMPC_OK;
}

template <int DP>
int launch_few_candidate_moments_dp(Handle* h, const GradArgs& g, const CostSlice& cs, size_t lds_bytes, hipStream_t s) {
    if (g.NXP == 1) return launch_few_candidate_moments<DP, 1>(h, g, cs, lds_bytes, s);
    return g.NXP == 2 ? launch_few_candidate_moments<DP, 2>(h, g, cs, lds_bytes, s) : launch_few_candidate_moments<DP, 6>(h, g, cs, lds_bytes, s);
}

template <int DP, int NT>
int launch_sweep(Handle* h, const GradArgs& g, size_t lds_bytes, hipStream_t s) {
    auto kern = (g.D == DP) ? adjoint_sweep_kernel<DP, NT, DP> : adjoint_sweep_kernel<DP, NT, 0>;
    int rc = allow_full_lds(h, reinterpret_cast<const void*>(kern));
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(g.B), dim3(NT), lds_bytes, s, g);
    GPMPC_HIP_CHECK(h, hipGetLastError());
    return GPMPC_OK;
}

template <int DP, int NXP>
int launch_moments_stream(Handle* h, const GradArgs& g, size_t lds_bytes, hipStream_t s) {
    auto kern = pair_moments_stream_kernel<DP, NXP>;
    int rc = allow_full_lds(h, reinterpret_cast<const void*>(kern));
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(g.H, g.B, g.gz), dim3(kGsThreads), lds_bytes, s, g);
    GPMPC_HIP_CHECK(h, hipGetLastError());
    return GPMPC_OK;
}

template <int DP>
int launch_moments_stream_dp(Handle* h, const GradArgs& g, size_t lds_bytes, hipStream_t s) {
    if (g.NXP == 1) return launch_moments_stream<DP, 1>(h, g, lds_bytes, s);
    return g.NXP == 2 ? launch_moments_stream<DP, 2>(h, g, lds_bytes, s) : launch_moments_stream<DP, 6>(h, g, lds_bytes, s);
}

template <int DP>
int launch_moments_dp(Handle* h, const GradArgs& g, size_t lds_bytes, hipStream_t s) {
    if (g.NXP == 1) return launch_moments<DP, 1>(h, g, lds_bytes, s);
    return g.NXP == 2 ? launch_moments<DP, 2>(h, g, lds_bytes, s) : launch_moments<DP, 6>(h, g, lds_bytes, s);
}

}  // namespace

int launch_rollout_grad(Handle* h, RolloutArgs& a, double* grad_out, hipStream_t s) {
    const int N = a.N, D = a.D, A = a.A, E = a.E, H = a.H, B = a.B;
    const int NX = E - D, P = D * (D + 1) / 2;
    h->last_grad_path = 0;
    if (D > 8) { h->last_grad_path = 8; return launch_rollout_grad_wide(h, a, grad_out, s); }
    int DP = 0;
    for (int v : {2, 3, 4, 6, 8}) if (D <= v) { DP = v; break; }
    if (DP == 0 || NX > 6) { h->err = "gradient: supported for D <= 8 with A (+ time) <= 6, and for 8 < D <= 16"; return GPMPC_ERR_LIMIT; }
    const int NXP = NX <= 1 ? 1 : (NX <= 2 ? 2 : 6);
    const int RS = grad_row_stride(DP, NXP);
    const int NSP = 1 + DP + DP * (DP + 1) / 2 + NXP;

    GradArgs g;
    memset(&g, 0, sizeof g);
    // tiling of the pairwise pass: row chunks of CH rows (64 to start with; the LDS-resident pass re-plans with the schedule model's
    // answer once it is known which pairs it is left with), as many output pairs per group as the LDS holds
    // one column per lane (two need > 128 VGPRs, i.e. 8 waves instead of 16 per workgroup: measured slower at config 2,
    // 2.86 vs 2.10 ms, round 3 -- that instantiation is no longer built)
    const int cols = 1;
    const int NCU = (N + cols - 1) / cols;
    // D <= 4: the sweep is one wavefront's, its prologue (cost adjoints and state-independent algebra of all H steps) every
    // wavefront's of the launch -- 8 of them (512 threads: 256 VGPRs, the 1024-thread form spills at D = 4) while each candidate has a CU to itself, 4 up to four per CU, else the one
    const int sweep_nt = DP <= 4 ? (B <= h->num_cu ? 512 : (B <= 4 * h->num_cu ? 256 : 64)) : 256;
    int CH = 0, RC = 0, NR = 0, wpp = 0, G = 0, gz = 1;
    size_t mom_lds = 0;
    int pairs_left = P;              // pairs the LDS-resident pass works on (known after the separable / tile passes were dispatched)
    size_t lds_budget = (size_t)h->lds_limit;
    // pairs per group (as many as the LDS budget holds, 0: none) and the LDS bytes for row chunks of `chunk_rows`
    auto fit = [&](int chunk_rows, int npairs, size_t budget, size_t& bytes) {
        const int rc_ = (N + chunk_rows - 1) / chunk_rows, nr_ = rc_ * chunk_rows, wpp_ = (rc_ * NCU + 63) / 64;
        for (int gg = npairs; gg >= 1; --gg) {
            const MomLayout L = make_mom_layout(N, D, E, gg, RS, nr_, wpp_, NSP);
            if ((size_t)L.total * 8 <= budget) { bytes = (size_t)L.total * 8; return gg; }
        }
        bytes = 0;
        return 0;
    };
    auto plan = [&](int chunk_rows) {
        CH = chunk_rows; RC = (N + CH - 1) / CH; NR = RC * CH;
        wpp = (RC * NCU + 63) / 64;
        gz = 1;
        G = fit(CH, pairs_left, lds_budget, mom_lds);
        // a small batch leaves most CUs idle: spread the pair groups of each (candidate, step) over up to P workgroups
        if (G > 0 && (long long)B * H * 2 <= h->num_cu) {
            int zmax = h->num_cu / (B * H);
            if (zmax > P) zmax = P;
            const int Gs = (P + zmax - 1) / zmax;            // pairs per workgroup
            if (Gs < G) { G = Gs; const MomLayout L = make_mom_layout(N, D, E, G, RS, NR, wpp, NSP); mom_lds = (size_t)L.total * 8; }
            gz = (P + G - 1) / G;
        }
        return G > 0 && (unsigned long long)RC * NCU * NCU < 0x100000000ULL && (unsigned long long)G * wpp * wpp < 0x100000000ULL;
    };
    const int CH0 = (N >= 64) ? 64 : ((N + 3) & ~3);
    plan(CH0);
    int pre_steps = 0;
    if (DP <= 4) {               // state-independent small algebra of all steps up front when it fits beside the rest
        const SweepLayout Lp = make_sweep_layout(D, A, E, H, NSP, sweep_nt / 64, 0, H);
        if ((size_t)Lp.total * 8 <= 96 * 1024) pre_steps = H;
    }
    const SweepLayout SL = make_sweep_layout(D, A, E, H, NSP, sweep_nt / 64, DP <= 4 ? 0 : kSweepAug, pre_steps);
    if ((size_t)SL.total * 8 > (size_t)h->lds_limit) { h->err = "gradient: horizon too long for the reverse sweep's LDS"; return GPMPC_ERR_LIMIT; }
    // memories whose per-point arrays do not fit the LDS (or on request) take the streaming moment pass
    bool stream = (G == 0) || h->opt_grad_stream == 1;
    size_t gs_lds = 0;
    if (stream) {
        const GsLayout GL = make_gs_layout(N, D, E, DP, NXP, RS, NSP);
        gs_lds = (size_t)GL.total * 8;
        if (gs_lds > (size_t)h->lds_limit) { h->err = "gradient: N too large for the column-factor array of the streaming moment pass"; return GPMPC_ERR_LIMIT; }
        gz = 1;
        if ((long long)B * H * 2 <= h->num_cu) { gz = h->num_cu / (B * H); if (gz > P) gz = P; }
    } else if ((unsigned long long)RC * NCU * NCU >= 0x100000000ULL || (unsigned long long)G * wpp * wpp >= 0x100000000ULL) {
        h->err = "gradient: index range too large for the multiply-high division"; return GPMPC_ERR_LIMIT;
    }
    auto magic = [](unsigned d) -> unsigned { return d <= 1 ? 0u : (unsigned)((0x100000000ULL + d - 1) / d); };

    // workspace: moments, mean sums, cost variances (when the caller does not keep them), flags of the separable pass
    const size_t n_mom = (size_t)B * H * P * NSP, n_ms = (size_t)B * H * D * mean_moment_count(D, NX), n_cv = (size_t)B * (H + 1);
    const size_t n_flag = ((size_t)B * H * P + 1) / 2;
    int rc = grow(h, h->gradws, n_mom + n_ms + n_cv + n_flag);
    if (rc) return rc;
    g.mom = h->gradws.p;
    g.msum = g.mom + n_mom;
    int* sep_flags = reinterpret_cast<int*>(g.msum + n_ms + n_cv);
    if (!a.cv_out) a.cv_out = g.msum + n_ms;

    // Diagonal pairs batch-major (pair_tile_grad_kernel.h) once the tables T_a are large and the batch fills the chip.  When the
    // forward itself takes the batch-major path, its tile pass forms these moments on the way (the same E_ij would otherwise be
    // evaluated twice: once for the forward's sums, once for the moments) -- the flags are then written DURING the forward.
    const bool want_tiles = h->opt_grad_tiles != 0 && D >= 2 && D <= 4 && tile_moments_supported(h, a, NSP) &&
                            (h->opt_grad_tiles == 2 || (4.0 * D * (double)N * N >= 6e6 && (long long)B * H >= 2LL * h->num_cu));
    if (want_tiles && tile_moments_fusable(h, a)) {
        GPMPC_HIP_CHECK(h, hipMemsetAsync(sep_flags, 0, (size_t)B * H * P * sizeof(int), s));
        a.grad_mom = g.mom; a.grad_done = sep_flags; a.grad_NSP = NSP; a.grad_NXP = NXP;
    }
    // few candidates: the stage costs / objective ride in the moment launch below (few_candidate_moments_kernel)
    const bool few = DP <= 4 && !stream && h->opt_grad_mean != 0 && h->opt_grad_merge != 0 && (long long)B * H * 2 <= h->num_cu &&
                     mom_lds >= 64 * sizeof(double);
    double* want_cm = a.cm_out; double* want_J = a.J_out;
    a.defer_cost = few ? 1 : 0;
    rc = launch_rollout(h, a, s);            // forward: trajectory, costs, J (+ the fused tile moments)
#if defined(GPMPC_HOST_TIMING)
    { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); g_host_timing_fwd = ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }
#endif
    a.defer_cost = 0;
    if (rc) return rc;
    bool cost_pending = few;
    const bool fused = h->last_fused_tiles != 0;

    g.Xt = a.Xt; g.beta = a.beta; g.Tm = a.Tm; g.ils2 = a.ils2; g.var = a.var; g.logvar = a.logvar; g.cost = a.cost;
    g.kappa = a.kappa; g.use_constraints = a.use_constraints;
    g.actions = a.actions; g.mu = a.mu_out; g.Sig = a.Sig_out; g.cv = a.cv_out;
    g.N = N; g.D = D; g.A = A; g.E = E; g.H = H; g.B = B; g.include_time = a.include_time; g.time0 = a.time0;
    g.grad = grad_out;
    g.DP = DP; g.NXP = NXP; g.NSP = NSP;
    g.pre_steps = pre_steps;
    g.cols = cols;
    g.host_out = h->hx_out; g.host_src = h->hx_src; g.host_n = (B == 1) ? h->hx_n : 0;
    g.host_flag = h->hio_flag_dev; g.host_flag_value = h->hio_seq;
    auto publish_plan = [&]() {
        g.G = G; g.CH = CH; g.RC = RC; g.wpp = wpp; g.gz = gz;
        g.magic_N = magic((unsigned)NCU); g.magic_wpp = magic((unsigned)wpp);
    };
    publish_plan();

    g.xrange = h->xrange.p; g.force_path = h->opt_force_path;
    // Off-diagonal pairs in separable form on the matrix cores (grad_sep_kernel.h) where the Taylor degree allows; the
    // element-wise kernels below skip the pairs it wrote.  Monomial tables: those of the forward kernel (ensure_monomials).
    g.sepdone = nullptr;
    // Measured (round 3, objective + gradient per launch, separable / element-wise): config 1 (N = 50, B = 256) 0.77 / 0.66 ms,
    // config 2 (N = 200) B = 256: 2.22 / 2.57 ms, B = 1: 0.65 / 0.61 ms (25 workgroups: the element-wise pass spreads its pair
    // groups over the idle CUs), config 3 (N = 500, B = 1024) 18.9 / 24.8 ms, config 4 (N = 1000, B = 2048): 69 ms for the 6
    // off-diagonal pairs against ~650 ms.  Hence from N = 128 on when the items fill the chip (option "grad_separable": 0 never,
    // 1 auto, 2 always).
    if (h->opt_grad_sep != 0 && ((N >= 128 && (long long)B * H >= h->num_cu) || h->opt_grad_sep == 2) && D >= 2 && D <= 4 && h->mono_D == D && h->mono_CM > 0 &&
        h->opt_force_path == 0) {
        int kmax = 0;
        for (int k = 1; k <= h->sep_kmax && k <= kMaxTaylor; ++k)
            if (h->mono_cum[k] <= 16 * kSepGradBlocks) kmax = k;
        const int nWt = 1 + D + D * (D + 1) / 2 + NX;
        if (kmax > 0 && nWt <= 32) {
            SepGradArgs sg;
            memset(&sg, 0, sizeof sg);
            sg.Xt = a.Xt; sg.beta = a.beta; sg.ils2 = a.ils2; sg.logvar = a.logvar; sg.xrange = h->xrange.p; sg.actions = a.actions;
            sg.mu = a.mu_out; sg.Sig = a.Sig_out; sg.mono_exp = h->mono_exp; sg.mono_w = h->mono_w.p;
            sg.mom = g.mom; sg.done = sep_flags;
            for (int k = 0; k < 16; ++k) sg.mono_cum[k] = h->mono_cum[k];
            sg.N = N; sg.D = D; sg.A = A; sg.E = E; sg.H = H; sg.B = B; sg.include_time = a.include_time; sg.time0 = a.time0;
            sg.NSP = NSP; sg.NXP = NXP; sg.kmax = kmax; sg.force_path = h->opt_force_path;
            sg.keep_diag_flags = fused ? 1 : 0;          // the fused forward already wrote the diagonal pairs' flags
            // weightings beyond a multiple of 16 (one or two) are accumulated on the VALU instead of opening another A block
            const int NE = (nWt > 16 && nWt % 16 != 0 && nWt % 16 <= 2) ? nWt % 16 : 0;
            const int NA = NE ? nWt / 16 : (nWt + 15) / 16;
            const bool v2 = sep_grad_version(DP) == 2;
            sg.PS = v2 ? sep_grad_point_words(D, 16 * NA + NE, kmax) : sep_grad_point_words_v1(D, NX, kmax);
            const int chunk = v2 ? sep_grad_chunk(DP) : 64;
            const int mat_words = nWt * 16 * kSepGradBlocks;          // moment matrix of a (pair, side): weightings x monomial slots
            sg.wave_words = chunk * sg.PS > mat_words ? chunk * sg.PS : mat_words;
            const int Poff = P - D;
            const size_t lds = ((size_t)rnd2(E) + rnd2(D * E) + rnd2((Poff > 0 ? Poff : 1) * DP * DP) + 64 + 8 + 16 * kSepGradBlocks
                                + 8 * kSepGradBlocks + 8 + (size_t)sep_grad_waves(DP) * sg.wave_words) * sizeof(double);
            if (lds <= (size_t)h->lds_limit) {
                auto launch = [&](auto kern) -> int {
                    int r2 = allow_full_lds(h, reinterpret_cast<const void*>(kern));
                    if (r2) return r2;
                    hipLaunchKernelGGL(kern, dim3(H, B), dim3(64 * sep_grad_waves(DP)), lds, s, sg);
                    GPMPC_HIP_CHECK(h, hipGetLastError());
                    return GPMPC_OK;
                };
                if (DP == 2) rc = launch(sep_grad_moments_kernel<2, 1>);
                else if (DP == 3) rc = launch(sep_grad_moments_kernel<3, 1>);
                else if (NE) rc = (NE == 1) ? launch(sep_grad_moments_kernel<4, 1, 1>) : launch(sep_grad_moments_kernel<4, 1, 2>);
                else rc = (NA == 1) ? launch(sep_grad_moments_kernel<4, 1>) : launch(sep_grad_moments_kernel<4, 2>);
                if (rc) return rc;
                g.sepdone = sep_flags;
                h->last_grad_path |= 1;
            }
        }
    }
    // Diagonal pairs batch-major over all (candidate, step) items of the stored trajectory when the forward did not form them;
    // the element-wise pass below keeps the mean sums and whatever is flagged 0.
    if (fused) {
        g.sepdone = sep_flags;
        h->last_grad_path |= 2 | 16;
    } else if (want_tiles) {
        if (!g.sepdone) {
            GPMPC_HIP_CHECK(h, hipMemsetAsync(sep_flags, 0, (size_t)B * H * P * sizeof(int), s));
            g.sepdone = sep_flags;
        }
        rc = launch_tile_moments(h, a, g.mom, sep_flags, NSP, NXP, s);
        if (rc) return rc;
        h->last_grad_path |= 2;
    }
    const bool merged = cost_pending && g.sepdone == nullptr && !want_tiles && gz >= 1;
    if (cost_pending && !merged) {
        // (the separable / tile passes took pairs after all: the costs as their own launch)
        rc = launch_traj_cost(h, a, want_cm, a.cv_out, want_J, s);
        if (rc) return rc;
        cost_pending = false;
    }
    if (DP <= 4 && h->opt_grad_mean != 0 && !merged) {
        // the mean part on its own, lanes over points (mean_moments_kernel): 22 -> ~1 ms of a config-4 launch (streaming pass); in the
        // LDS-resident pass it was a quarter of the kernel's time per (candidate, step) at config 2 (profiles/r04j_moment_phases.txt)
        auto launch = [&](auto kern) -> int {
            hipLaunchKernelGGL(kern, dim3(H, B), dim3(64 * DP), 0, s, g);
            GPMPC_HIP_CHECK(h, hipGetLastError());
            return GPMPC_OK;
        };
        if (DP == 2) rc = NXP == 1 ? launch(mean_moments_kernel<2, 1>) : (NXP == 2 ? launch(mean_moments_kernel<2, 2>) : launch(mean_moments_kernel<2, 6>));
        else if (DP == 3) rc = NXP == 1 ? launch(mean_moments_kernel<3, 1>) : (NXP == 2 ? launch(mean_moments_kernel<3, 2>) : launch(mean_moments_kernel<3, 6>));
        else rc = NXP == 1 ? launch(mean_moments_kernel<4, 1>) : (NXP == 2 ? launch(mean_moments_kernel<4, 2>) : launch(mean_moments_kernel<4, 6>));
        if (rc) return rc;
        g.mean_done = 1;
        h->last_grad_path |= 32;
    }
    if (stream) {
        h->last_grad_path |= 4;
        switch (DP) {
            case 2:  rc = launch_moments_stream_dp<2>(h, g, gs_lds, s); break;
            case 3:  rc = launch_moments_stream_dp<3>(h, g, gs_lds, s); break;
            case 4:  rc = launch_moments_stream_dp<4>(h, g, gs_lds, s); break;
            case 6:  rc = launch_moments_stream_dp<6>(h, g, gs_lds, s); break;
            default: rc = launch_moments_stream_dp<8>(h, g, gs_lds, s); break;
        }
    } else {
        int left = 0;
        for (int a1 = 0; a1 < D; ++a1)
            for (int b1 = a1; b1 < D; ++b1)
                left += ((a1 == b1) ? (h->last_grad_path & 2) != 0 : (h->last_grad_path & 1) != 0) ? 0 : 1;
        pairs_left = left > 0 ? left : 1;
        // Two workgroups per CU, half the LDS each (option "grad_share_cu": 0 auto, 1 wherever it fits, 2 never).  Measured
        // (profiles/r04j_grad_sweep.txt, objective + gradient per launch): config 2 B = 256 2.04 -> 1.84 ms, B = 1024 7.57 -> 7.01,
        // config 1 B = 2048 3.58 -> 2.92, config 3 B = 1024 18.2 -> 17.7; nothing at B = 1 -- hence from two workgroups per CU on.
        bool share = DP <= 3 && cols == 1 && h->opt_grad_share != 2 && (h->opt_grad_share == 1 || (long long)B * H >= 2LL * h->num_cu);
        if (share) {
            lds_budget = (size_t)h->lds_limit / 2;
            if (!plan(CH0)) { share = false; lds_budget = (size_t)h->lds_limit; }
        }
        g.share_cu = share ? 1 : 0;
        // Row-chunk length from the schedule model (moment_schedule.h), as a function of the MODEL'S SHAPE ONLY -- evaluated for the
        // throughput configuration (the pairs the element-wise pass keeps when the separable pass takes the off-diagonal ones,
        // two workgroups per CU where they fit, no spreading over blockIdx.z) whatever the batch at hand: every pair's sums are
        // then formed in the same order for any batch size, grouping and workgroup shape (the lockstep L-BFGS restarts rely on
        // one candidate's gradient being bit-identical alone and inside a batch).
        int want = CH0;
        if (h->opt_grad_chunk > 0) want = h->opt_grad_chunk < CH0 ? h->opt_grad_chunk : CH0;
        else {
            const bool sep_shape = h->opt_grad_sep != 0 && N >= 128 && D >= 2 && D <= 4 && h->opt_force_path == 0;
            std::vector<int> pair_is_diag;
            for (int a1 = 0; a1 < D; ++a1)
                for (int b1 = a1; b1 < D; ++b1)
                    if (a1 == b1 || !sep_shape) pair_is_diag.push_back(a1 == b1 ? 1 : 0);
            const int npairs = (int)pair_is_diag.size();
            size_t bytes = 0;
            const bool two = DP <= 3 && cols == 1 && h->opt_grad_share != 2 && fit(CH0, npairs, (size_t)h->lds_limit / 2, bytes) > 0;
            const size_t budget = two ? (size_t)h->lds_limit / 2 : (size_t)h->lds_limit;
            const int NW = (cols == 2 || two) ? 8 : (DP <= 3 ? 16 : 8);
            const int key[8] = {N, D, E, cols, NW, (int)(budget >> 10), npairs, sep_shape ? 1 : 0};
            if (memcmp(key, h->chunk_key, sizeof key) == 0 && h->chunk_rows > 0) want = h->chunk_rows;
            else {
                want = choose_moment_chunk(N, cols, NW, CH0, pair_is_diag, [&](int c, int& Gc, int& gzc) {
                    size_t bb = 0;
                    Gc = fit(c, npairs, budget, bb);
                    gzc = 1;
                    return Gc > 0;
                });
                memcpy(h->chunk_key, key, sizeof key);
                h->chunk_rows = want;
            }
        }
        if (!plan(want) && share) {                 // the chosen chunk does not fit twice: one workgroup per CU rather than another chunk length
            share = false;
            g.share_cu = 0;
            lds_budget = (size_t)h->lds_limit;
        }
        if (!plan(want)) plan(CH0);
        publish_plan();
    if (merged) {
        g.mean_done = 1;
        h->last_grad_path |= 32 | 64;
        const CostSlice cs{a.cost, a.kappa, a.clip, a.use_constraints, want_cm, a.cv_out, want_J};
        switch (DP) {
            case 2:  rc = launch_few_candidate_moments_dp<2>(h, g, cs, mom_lds, s); break;
            case 3:  rc = launch_few_candidate_moments_dp<3>(h, g, cs, mom_lds, s); break;
            default: rc = launch_few_candidate_moments_dp<4>(h, g, cs, mom_lds, s); break;
        }
    } else
    switch (DP) {
        case 2:  rc = launch_moments_dp<2>(h, g, mom_lds, s); break;
        case 3:  rc = launch_moments_dp<3>(h, g, mom_lds, s); break;
        case 4:  rc = launch_moments_dp<4>(h, g, mom_lds, s); break;
        case 6:  rc = launch_moments_dp<6>(h, g, mom_lds, s); break;
        default: rc = launch_moments_dp<8>(h, g, mom_lds, s); break;
    }
    }
    if (rc) return rc;
    switch (DP) {
        case 2:  rc = sweep_nt == 512 ? launch_sweep<2, 512>(h, g, (size_t)SL.total * 8, s) : (sweep_nt == 256 ? launch_sweep<2, 256>(h, g, (size_t)SL.total * 8, s) : launch_sweep<2, 64>(h, g, (size_t)SL.total * 8, s)); break;
        case 3:  rc = sweep_nt == 512 ? launch_sweep<3, 512>(h, g, (size_t)SL.total * 8, s) : (sweep_nt == 256 ? launch_sweep<3, 256>(h, g, (size_t)SL.total * 8, s) : launch_sweep<3, 64>(h, g, (size_t)SL.total * 8, s)); break;
        case 4:  rc = sweep_nt == 512 ? launch_sweep<4, 512>(h, g, (size_t)SL.total * 8, s) : (sweep_nt == 256 ? launch_sweep<4, 256>(h, g, (size_t)SL.total * 8, s) : launch_sweep<4, 64>(h, g, (size_t)SL.total * 8, s)); break;
        case 6:  rc = launch_sweep<6, 256>(h, g, (size_t)SL.total * 8, s); break;
        default: rc = launch_sweep<8, 256>(h, g, (size_t)SL.total * 8, s); break;
    }
    if (rc) return rc;
    return GPMPC_OK;
}

}  // namespace gpmpc_hip
