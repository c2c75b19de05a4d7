// rollout_kernel.h -- the GP-MPC inner loop for gfx950 (MI355X, CDNA4).
//
// One workgroup owns one candidate action sequence for the whole horizon: the H-step
// recurrence (reference rl_gp_mpc/control_objects/models/gp_model.py:60-110) is sequential
// per candidate but candidates are independent, so no inter-workgroup traffic exists and the
// whole batch is ONE launch.  Per horizon step the workgroup evaluates the moment-matched GP
// prediction (gp_model.py:112-180) in a restructured, memory-lean form:
//
//   input covariance is non-zero only in its state block, so every E x E solve of the
//   reference collapses to a D x D one:
//     mean   : q_ai = nu_i^T (Sigma + Lambda_a)^-1 nu_i           (= iN B^-1 iN^T, :140-148)
//     cov    : L_ab,ij = exp(ka'_i + kb'_j + g_i . w_j)            (= exp(k_a + k_b + maha), :161-169)
//              with Z = R_ab^-1 Sigma (= 2Q, :163), u_i = nu_i/l_a^2, w_j = nu_j/l_b^2,
//              ka'_i = k_ai + u_i^T Z u_i / 2, kb'_j = k_bj + w_j^T Z w_j / 2, g_i = Z u_i
//     S_ab   = [sum_ij beta_ai L_ab,ij beta_bj - d_ab sum_ij iK_a,ij L_aa,ij]/sqrt|R_ab| ...   (:170-178)
//   The (D,D,N,N) tensors of the reference (:166,169-171) never exist: the N x N pairwise
//   work of each output pair a <= b is streamed through registers; for a == b the two sums
//   are merged through T_a = beta_a beta_a^T - iK_a and only i <= j is visited (L_aa and T_a
//   are symmetric).
//
// Mapping to the hardware: lanes of a 64-wide wavefront own consecutive columns j (coalesced
// T_a reads, conflict-free LDS reads), the row operands (ka', beta_a, g_i) are LDS broadcasts,
// waves pull (pair, row-chunk, 64-column) work items from an LDS counter so that the
// triangular diagonal pairs balance, and every reduction is a fixed-order wavefront butterfly
// followed by a fixed-order sum, so results are bitwise reproducible run to run.
#pragma once
#include "gpmpc_internal.h"

#ifdef GPMPC_TRACE_ON
#define GPMPC_TRACE(id) do { if (threadIdx.x == 0 && blockIdx.x == 0) printf("trace %d t=%d\n", id, t_dbg); } while (0)
#else
#define GPMPC_TRACE(id) do {} while (0)
#endif

namespace gpmpc_hip {

// ------------------------------------------------------------------------------------------
// LDS / scratch layout (offsets in doubles), shared by host (sizing) and device (carving).
struct Layout {
    int mu, Sig, m, M, cc, s1, Vs, Sp, TS, v1, v2, ev, misc, rdet, aug, part, ints;
    int lds_total;     // doubles of LDS
    // per-point arrays: in LDS (offsets from smem) or in global scratch (offsets from base)
    int nu, kk, lb, rows, kb;
    int pp_total;      // doubles of the per-point block
};

__host__ __device__ inline int rnd2(int x) { return (x + 1) & ~1; }

__host__ __device__ inline Layout make_layout(int N, int D, int A, int E, int G, int DP, int wpp,
                                              bool global_scratch) {
    Layout L;
    const int P = D * (D + 1) / 2;
    int o = 0;
    L.mu = o;   o += rnd2(D);
    L.Sig = o;  o += rnd2(D * D);
    L.m = o;    o += rnd2(E);
    L.M = o;    o += rnd2(D);
    L.cc = o;   o += rnd2(D);
    L.s1 = o;   o += rnd2(D * (D + 1));
    L.Vs = o;   o += rnd2(D * D);
    L.Sp = o;   o += rnd2(P);
    L.TS = o;   o += rnd2(D * D);
    L.v1 = o;   o += rnd2(D);
    L.v2 = o;   o += rnd2(D);
    L.ev = o;   o += rnd2(D + A);
    L.misc = o; o += 8;
    L.rdet = o; o += rnd2(G);
    const int nprob = D > G ? D : G;
    L.aug = o;  o += nprob * 2 * D * D;
    L.part = o; o += rnd2(G * wpp);
    L.ints = o; o += rnd2((2 * P + 4 + 1) / 2);   // pair tables pa[P], pb[P] + counter (ints)
    int q = global_scratch ? 0 : o;
    L.nu = q;   q += rnd2(D * N);
    L.kk = q;   q += rnd2(D * N);
    L.lb = q;   q += rnd2(D * N);
    L.rows = q; q += G * N * (DP + 2);
    L.kb = q;   q += rnd2(G * N);
    if (global_scratch) { L.lds_total = o; L.pp_total = q; }
    else                { L.lds_total = q; L.pp_total = q - o; }
    return L;
}

// ------------------------------------------------------------------------------------------
__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Gaussian elimination with partial pivoting on an augmented [A | RHS] block (row stride ld).
// The solution replaces the RHS; returns det(A).  Same algorithm class (LU, partial pivoting)
// as the torch.linalg.solve / torch.det calls of the reference (gp_model.py:146,150,163,176).
__device__ inline double gauss_solve(double* aug, int D, int nrhs, int ld) {
    double det = 1.0;
    const int nc = D + nrhs;
    for (int k = 0; k < D; ++k) {
        int piv = k;
        double best = fabs(aug[k * ld + k]);
        for (int r = k + 1; r < D; ++r) {
            double v = fabs(aug[r * ld + k]);
            if (v > best) { best = v; piv = r; }
        }
        if (piv != k) {
            for (int c = k; c < nc; ++c) {
                double t = aug[k * ld + c];
                aug[k * ld + c] = aug[piv * ld + c];
                aug[piv * ld + c] = t;
            }
            det = -det;
        }
        const double pv = aug[k * ld + k];
        det *= pv;
        const double ip = 1.0 / pv;
        for (int r = k + 1; r < D; ++r) {
            const double f = aug[r * ld + k] * ip;
            for (int c = k + 1; c < nc; ++c) aug[r * ld + c] -= f * aug[k * ld + c];
        }
    }
    for (int k = D - 1; k >= 0; --k) {
        const double ip = 1.0 / aug[k * ld + k];
        for (int c = D; c < nc; ++c) {
            double s = aug[k * ld + c];
            for (int r = k + 1; r < D; ++r) s -= aug[k * ld + r] * aug[r * ld + c];
            aug[k * ld + c] = s * ip;
        }
    }
    return det;
}

__device__ inline double norm_cdf_ref(double x, double mu, double sigma) {
    // normal_cdf of the reference (control_objects/utils/pytorch_utils.py:16-17)
    return 0.5 * (1.0 + erf((x - mu) / (sigma * 1.4142135623730951)));
}

// ------------------------------------------------------------------------------------------
template <int DP, int NT, bool GLOBAL>
__global__ __launch_bounds__(NT) void rollout_kernel(const RolloutArgs p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int NW = NT / kWave;
    constexpr int RS = DP + 2;              // row record: ka', beta_a, g[0..DP)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int c = blockIdx.x;
    const int N = p.N, D = p.D, A = p.A, E = p.E, H = p.H, G = p.G;
    const int P = D * (D + 1) / 2;
    const int DA = D + A;
    const int LD = 2 * D;                   // row stride of an augmented block
    const int wpp = (p.RC * N + 63) / 64;   // wave items per output pair

    const Layout L = make_layout(N, D, A, E, G, DP, wpp, GLOBAL);
    double* s_mu = smem + L.mu;
    double* s_Sig = smem + L.Sig;
    double* s_m = smem + L.m;
    double* s_M = smem + L.M;
    double* s_cc = smem + L.cc;
    double* s_s1 = smem + L.s1;             // [a][0] = sum lb, [a][1+d] = sum lb * nu_d
    double* s_Vs = smem + L.Vs;             // [k][a]
    double* s_Sp = smem + L.Sp;
    double* s_TS = smem + L.TS;
    double* s_v1 = smem + L.v1;
    double* s_v2 = smem + L.v2;
    double* s_ev = smem + L.ev;
    double* s_misc = smem + L.misc;         // [0] = running J
    double* s_rdet = smem + L.rdet;
    double* s_aug = smem + L.aug;
    double* s_part = smem + L.part;
    int* s_pa = reinterpret_cast<int*>(smem + L.ints);
    int* s_pb = s_pa + P;
    int* s_counter = s_pb + P;

    double* ppbase = GLOBAL ? (p.scratch + (size_t)c * p.scratch_stride) : smem;
    double* a_nu = ppbase + L.nu;           // [d][p]
    double* a_kk = ppbase + L.kk;           // [a][p]
    double* a_lb = ppbase + L.lb;           // [a][p]
    double* a_rows = ppbase + L.rows;       // [gq][p][RS]
    double* a_kb = ppbase + L.kb;           // [gq][p]

    const double* cost_target = p.cost;
    const double* cost_W = p.cost + DA;
    const double* cost_WT = cost_W + DA * DA;
    const double* cost_smin = cost_WT + D * D;
    const double* cost_smax = cost_smin + D;
    const double* act = p.actions + (size_t)c * H * A;

    [[maybe_unused]] int t_dbg = -1;
    // ---- init -----------------------------------------------------------------------
    for (int i = tid; i < D; i += NT) s_mu[i] = p.mu0[i];
    for (int i = tid; i < D * D; i += NT) s_Sig[i] = p.S0[i];
    if (tid == 0) {
        s_misc[0] = 0.0;
        int q = 0;
        for (int a = 0; a < D; ++a)
            for (int b = a; b < D; ++b) { s_pa[q] = a; s_pb[q] = b; ++q; }
    }
    __syncthreads();
    GPMPC_TRACE(1);
    if (p.mu_out)
        for (int i = tid; i < D; i += NT) p.mu_out[((size_t)c * (H + 1)) * D + i] = s_mu[i];
    if (p.Sig_out)
        for (int i = tid; i < D * D; i += NT) p.Sig_out[((size_t)c * (H + 1)) * D * D + i] = s_Sig[i];

    for (int t = 0; t <= H; ++t) {
        t_dbg = t;
        const bool terminal = (t == H);
        // ---- stage / terminal cost of (mu_t, Sigma_t, a_t) ------------------------------
        // setpoint_distance_reward_mapper.py:36-56 (stage), :135-141 (terminal)
        {
            const int n = terminal ? D : DA;
            const double* Wm = terminal ? cost_WT : cost_W;
            for (int i = tid; i < n; i += NT)
                s_ev[i] = (i < D ? s_mu[i] : act[t * A + (i - D)]) - cost_target[i];
            if (!terminal) {
                // input mean of this step: [mu, a_t, (time)]  (gp_model.py:98-102)
                for (int i = tid; i < E; i += NT) {
                    double v;
                    if (i < D) v = s_mu[i];
                    else if (i < DA) v = act[t * A + (i - D)];
                    else v = p.time0 + (double)t;
                    s_m[i] = v;
                }
            }
            __syncthreads();
            GPMPC_TRACE(2);
            for (int idx = tid; idx < D * D; idx += NT) {
                const int i = idx / D, j = idx - i * D;
                double s = 0.0;
                for (int k = 0; k < D; ++k) s = fma(Wm[i * n + k], s_Sig[k * D + j], s);
                s_TS[idx] = s;                                   // TS = W Sigma  (:52 / :138)
            }
            for (int k = tid; k < 2 * D; k += NT) {
                double s = 0.0;
                if (k < D) { for (int i = 0; i < n; ++i) s = fma(s_ev[i], Wm[i * n + k], s); s_v1[k] = s; }
                else { const int kk = k - D; for (int j = 0; j < n; ++j) s = fma(Wm[kk * n + j], s_ev[j], s); s_v2[kk] = s; }
            }
            __syncthreads();
            GPMPC_TRACE(3);
            if (wave == NW - 1) {
                double cm = 0.0, cv = 0.0;
                for (int idx = lane; idx < D * D; idx += 64) {
                    const int i = idx / D, j = idx - i * D;
                    cm = fma(s_Sig[idx], Wm[j * n + i], cm);                       // tr(Sigma W)
                    cv = fma(2.0 * s_TS[idx], s_TS[j * D + i], cv);                // tr(2 TS TS)
                    cv = fma(4.0 * s_v1[i] * s_Sig[idx], s_v2[j], cv);             // 4 e^T TS W e
                }
                for (int idx = lane; idx < n * n; idx += 64) {
                    const int i = idx / n, j = idx - i * n;
                    cm = fma(s_ev[i] * Wm[idx], s_ev[j], cm);                      // e^T W e
                }
                if (p.use_constraints && !terminal) {                               // :58-66
                    for (int d = lane; d < D; d += 64) {
                        const double sg = s_Sig[d * D + d];   // reference passes the VARIANCE as sigma
                        cm += norm_cdf_ref(cost_smin[d], s_mu[d], sg) + (1.0 - norm_cdf_ref(cost_smax[d], s_mu[d], sg));
                    }
                }
                cm = wave_sum(cm);
                cv = wave_sum(cv);
                if (lane == 0) {
                    double ucb = -cm + p.kappa * sqrt(cv);                          // gp_mpc_controller.py:270
                    if (p.clip) ucb = fmin(ucb, 0.0);                               // :272-274
                    s_misc[0] -= ucb;
                    if (p.cm_out) p.cm_out[(size_t)c * (H + 1) + t] = cm;
                    if (p.cv_out) p.cv_out[(size_t)c * (H + 1) + t] = cv;
                }
            }
        }
        if (terminal) break;

        // ---- Phase A: mean part -------------------------------------------------------
        // A_a = Sigma + diag(l_a^2) -> A_a^-1, det  (restated B of gp_model.py:141)
        if (tid < D) {
            const int a = tid;
            double* aug = s_aug + a * (D * LD);
            double prodil = 1.0;
            for (int i = 0; i < D; ++i) {
                const double il2 = p.ils2[a * E + i];
                prodil *= il2;
                for (int j = 0; j < D; ++j) {
                    aug[i * LD + j] = s_Sig[i * D + j] + (i == j ? 1.0 / il2 : 0.0);
                    aug[i * LD + D + j] = (i == j ? 1.0 : 0.0);
                }
            }
            const double detA = gauss_solve(aug, D, D, LD);
            s_cc[a] = p.var[a] / sqrt(detA * prodil);            // c_a = var_a / sqrt(det B_a)  (:150)
        }
        __syncthreads();
        GPMPC_TRACE(4);
        // per point: nu, k_a, lb_a
        for (int it = tid; it < D * N; it += NT) {
            const int a = it / N, pt = it - a * N;
            double nu[DP];
#pragma unroll
            for (int d = 0; d < DP; ++d) nu[d] = (d < D) ? (p.Xt[d * N + pt] - s_m[d]) : 0.0;
            const double* Ai = s_aug + a * (D * LD) + D;          // A_a^-1 [i][j] at Ai[i*LD + j]
            double q = 0.0, ks = 0.0;
#pragma unroll
            for (int i = 0; i < DP; ++i) {
                if (i < D) {
                    double r = 0.0;
#pragma unroll
                    for (int j = 0; j < DP; ++j)
                        if (j < D) r = fma(Ai[i * LD + j], nu[j], r);
                    q = fma(nu[i], r, q);
                    ks = fma(nu[i] * nu[i], p.ils2[a * E + i], ks);
                }
            }
            for (int e = D; e < E; ++e) {
                const double v = p.Xt[e * N + pt] - s_m[e];
                const double w2 = v * v * p.ils2[a * E + e];
                q += w2;
                ks += w2;
            }
            a_kk[it] = p.logvar[a] - 0.5 * ks;                                       // k_a  (:168)
            a_lb[it] = exp(-0.5 * q) * p.beta[it];                                   // lb   (:148)
            if (a == 0) {
#pragma unroll
                for (int d = 0; d < DP; ++d)
                    if (d < D) a_nu[d * N + pt] = nu[d];
            }
        }
        __syncthreads();
        GPMPC_TRACE(5);
        // s1[a][0] = sum_p lb, s1[a][1+d] = sum_p lb nu_d   (fixed order: lane-strided + butterfly)
        for (int s = wave; s < D * (D + 1); s += NW) {
            const int a = s / (D + 1), dd = s - a * (D + 1);
            double v = 0.0;
            if (dd == 0) { for (int pt = lane; pt < N; pt += 64) v += a_lb[a * N + pt]; }
            else { for (int pt = lane; pt < N; pt += 64) v = fma(a_lb[a * N + pt], a_nu[(dd - 1) * N + pt], v); }
            v = wave_sum(v);
            if (lane == 0) s_s1[s] = v;
        }
        __syncthreads();
        GPMPC_TRACE(6);
        if (tid < D) s_M[tid] = s_cc[tid] * s_s1[tid * (D + 1)];                     // M_a (:152)
        for (int idx = tid; idx < D * D; idx += NT) {
            const int k = idx / D, a = idx - k * D;
            const double* Ai = s_aug + a * (D * LD) + D;
            double s = 0.0;
            for (int j = 0; j < D; ++j) s = fma(Ai[k * LD + j], s_s1[a * (D + 1) + 1 + j], s);
            s_Vs[idx] = s_cc[a] * s;                                                 // state rows of V (:153)
        }
        __syncthreads();
        GPMPC_TRACE(7);

        // ---- Phase B: covariance part, output pairs in groups of G ----------------------
        for (int q0 = 0; q0 < P; q0 += G) {
            const int Gc = (P - q0 < G) ? (P - q0) : G;
            if (tid < Gc) {
                const int a = s_pa[q0 + tid], b = s_pb[q0 + tid];
                double* aug = s_aug + tid * (D * LD);
                for (int i = 0; i < D; ++i)
                    for (int j = 0; j < D; ++j) {
                        const double dab = p.ils2[a * E + j] + p.ils2[b * E + j];
                        aug[i * LD + j] = s_Sig[i * D + j] * dab + (i == j ? 1.0 : 0.0);   // R (:156-159)
                        aug[i * LD + D + j] = s_Sig[i * D + j];
                    }
                const double detR = gauss_solve(aug, D, D, LD);                      // Z = R^-1 Sigma = 2Q (:163)
                s_rdet[tid] = 1.0 / sqrt(detR);                                      // (:176)
            }
            if (tid == NT - 1) *s_counter = 0;
            __syncthreads();
            GPMPC_TRACE(8);
            for (int it = tid; it < Gc * N; it += NT) {
                const int gq = it / N, pt = it - gq * N;
                const int a = s_pa[q0 + gq], b = s_pb[q0 + gq];
                const double* Z = s_aug + gq * (D * LD) + D;
                double u[DP], w[DP];
#pragma unroll
                for (int d = 0; d < DP; ++d) {
                    const double v = (d < D) ? a_nu[d * N + pt] : 0.0;
                    u[d] = (d < D) ? v * p.ils2[a * E + d] : 0.0;
                    w[d] = (d < D) ? v * p.ils2[b * E + d] : 0.0;
                }
                double qa = 0.0, qb = 0.0;
                double g[DP];
#pragma unroll
                for (int d = 0; d < DP; ++d) g[d] = 0.0;
                double* rec = a_rows + ((size_t)gq * N + pt) * RS;
#pragma unroll
                for (int i = 0; i < DP; ++i) {
                    if (i < D) {
                        double zu = 0.0, zw = 0.0;
#pragma unroll
                        for (int j = 0; j < DP; ++j)
                            if (j < D) {
                                const double z = Z[i * LD + j];
                                zu = fma(z, u[j], zu);
                                zw = fma(z, w[j], zw);
                                g[j] = fma(z, u[i], g[j]);      // g = Z^T u: cross term u^T Z w = g . w
                            }
                        qa = fma(u[i], zu, qa);
                        qb = fma(w[i], zw, qb);
                    }
                }
#pragma unroll
                for (int d = 0; d < DP; ++d) rec[2 + d] = g[d];
                rec[0] = a_kk[a * N + pt] + 0.5 * qa;
                rec[1] = p.beta[a * N + pt];
                a_kb[gq * N + pt] = a_kk[b * N + pt] + 0.5 * qb;
            }
            __syncthreads();
            GPMPC_TRACE(9);
            // pairwise N x N work: waves pull (pair, row chunk, 64 columns) items
            const int total = Gc * wpp;
            auto pull_item = [&]() -> int {
                int pulled = 0;
                if (lane == 0) pulled = __hip_atomic_fetch_add(s_counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                return __builtin_amdgcn_readfirstlane(pulled);       // wave-uniform (SGPR) work item
            };
            for (int wi = pull_item(); wi < total; wi = pull_item()) {
                const int gq = wi / wpp;
                const int flat = (wi - gq * wpp) * 64 + lane;
                const int a = __builtin_amdgcn_readfirstlane(s_pa[q0 + gq]);
                const int b = __builtin_amdgcn_readfirstlane(s_pb[q0 + gq]);
                const bool diag = (a == b);
                const bool valid = flat < p.RC * N;
                const int r = valid ? flat / N : 0;
                const int j = valid ? flat - r * N : 0;
                const int i0 = r * p.CH;
                int i1 = i0 + p.CH;
                if (i1 > N) i1 = N;
                if (diag && i1 > j + 1) i1 = j + 1;
                const int len = valid ? (i1 - i0) : 0;
                double acc = 0.0;
                if (len > 0) {
                    double w[DP];
#pragma unroll
                    for (int d = 0; d < DP; ++d) w[d] = (d < D) ? a_nu[d * N + j] * p.ils2[b * E + d] : 0.0;
                    const double kbj = a_kb[gq * N + j];
                    const double* rec = a_rows + ((size_t)gq * N + i0) * RS;
                    if (diag) {
                        const double* Tp = p.Tm + ((size_t)a * N + i0) * N + j;
                        for (int it = 0; it < len; ++it) {
                            double arg = rec[0] + kbj;
#pragma unroll
                            for (int d = 0; d < DP; ++d) arg = fma(rec[2 + d], w[d], arg);
                            acc = fma(exp(arg), *Tp, acc);
                            rec += RS;
                            Tp += N;
                        }
                        acc *= 2.0;
                    } else {
                        for (int it = 0; it < len; ++it) {
                            double arg = rec[0] + kbj;
#pragma unroll
                            for (int d = 0; d < DP; ++d) arg = fma(rec[2 + d], w[d], arg);
                            acc = fma(exp(arg), rec[1], acc);
                            rec += RS;
                        }
                        acc *= p.beta[b * N + j];
                    }
                }
                acc = wave_sum(acc);
                if (lane == 0) s_part[wi] = acc;
            }
            __syncthreads();
            GPMPC_TRACE(10);
            if (tid < Gc) {
                double s = 0.0;
                for (int k = 0; k < wpp; ++k) s += s_part[tid * wpp + k];
                s_Sp[q0 + tid] = s * s_rdet[tid];
            }
            __syncthreads();
            GPMPC_TRACE(11);
        }

        // ---- Phase C: state update  (gp_model.py:105-108, 177-178) -------------------------
        for (int idx = tid; idx < D * D; idx += NT) {
            const int i = idx / D, j = idx - i * D;
            const int a = i < j ? i : j, b = i < j ? j : i;
            const int q = a * D - (a * (a - 1)) / 2 + (b - a);
            double S = s_Sp[q] - s_M[i] * s_M[j] + (i == j ? p.var[i] : 0.0);
            double cij = 0.0, cji = 0.0;
            for (int k = 0; k < D; ++k) {
                cij = fma(s_Sig[i * D + k], s_Vs[k * D + j], cij);
                cji = fma(s_Sig[j * D + k], s_Vs[k * D + i], cji);
            }
            s_TS[idx] = S + s_Sig[idx] + (cij + cji);   // (cij + cji) commutes: Sigma stays exactly symmetric
        }
        __syncthreads();
        GPMPC_TRACE(12);
        for (int idx = tid; idx < D * D; idx += NT) {
            s_Sig[idx] = s_TS[idx];
            if (p.Sig_out) p.Sig_out[((size_t)c * (H + 1) + (t + 1)) * D * D + idx] = s_TS[idx];
        }
        for (int i = tid; i < D; i += NT) {
            const double v = s_mu[i] + s_M[i];
            s_mu[i] = v;
            if (p.mu_out) p.mu_out[((size_t)c * (H + 1) + (t + 1)) * D + i] = v;
        }
        __syncthreads();
        GPMPC_TRACE(13);
    }
    __syncthreads();
    GPMPC_TRACE(14);
    if (tid == (NW - 1) * 64 && p.J_out) p.J_out[c] = s_misc[0] / (double)(H + 1);     // mean over H+1 (:275-276)
}

}  // namespace gpmpc_hip
