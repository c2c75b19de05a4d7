// grad_kernels.h -- analytic gradient dJ/du of the mean-LCB objective (the reference obtains it by
// torch autograd through predict_trajectory: gp_mpc_controller.py:277 `mean_cost.backward()`, :285).
//
// Two kernels after the forward rollout has stored the trajectory (mu_t, Sigma_t):
//
//  pair_moments_kernel  one workgroup per (candidate, horizon step) -- the steps are independent once the
//      trajectory is known, so the N^2 work of the gradient is B*H-way parallel.  It re-evaluates the
//      pairwise weights E_ij = T_ij exp(ka'_i + kb'_j + g_i . w_j) of gp_model.py:161-175 exactly as the
//      forward kernel does and accumulates, per output pair (a, b), the moments of p_ij = u_i + w_j:
//          W = sum E_ij, P1 = sum E_ij p_ij, P2 = sum E_ij p_ij p_ij^T, Pe = sum E_ij (nu_ie/l_ae^2 + nu_je/l_be^2)
//      (all nu_i = x_i - m move together with the input mean, and Z = R^-1 Sigma enters the exponent as
//      1/2 p^T Z p, so dW/dm and dW/dZ are these moments).  Lanes own columns j: column sums and the
//      u_i-weighted column sums accumulate in registers, one wavefront reduction per work item.
//  adjoint_sweep_kernel one workgroup per candidate, t = H-1 .. 0: D x D algebra on the stored moments plus
//      one O(N D^2) pass over the points for the mean part (gp_model.py:140-153).
//
// The derivation is written out in DESIGN.md (section 4.4); a numpy statement of it lives with the tests.
#pragma once
#include "rollout_kernel.h"

namespace gpmpc_hip {

struct GradArgs {
    const double* Xt;      // (E, N)
    const double* beta;    // (D, N)
    const double* Tm;      // (D, N + kTPad, N)
    const double* ils2;    // (D, E)
    const double* var;     // (D)
    const double* logvar;  // (D)
    const double* cost;    // target | W | W_T | smin | smax
    double kappa;
    int use_constraints;
    const double* actions;  // (B, H, A)
    const double* mu;       // (B, H+1, D)     stored trajectory of the forward launch
    const double* Sig;      // (B, H+1, D, D)
    const double* cv;       // (B, H+1)        cost variances of the forward launch
    int N, D, A, E, H, B;
    int include_time;
    double time0;
    double* mom;     // (B, H, P, NSP)   [W | P1 (DP) | P2 upper triangle (DP (DP+1)/2) | Pe (NXP)]
    double* msum;    // (B, H, D, NM)    moments of nu under lb_a up to third order (mean_moment_count)
    double* grad;    // (B, H, A)
    const double* xrange;   // (2, E) min / max of the memory points per input dimension
    int pre_steps;          // sweep: H if the state-independent algebra of all steps fits the LDS (computed up front), else 0
    int DP, NXP, NSP;
    int force_path;         // 0 auto, 1 always the direct exp form (tests)
    int cols;               // columns per lane in the pairwise pass (1 or 2)
    int G, CH, RC, wpp;
    int gz;                 // workgroups per (candidate, step) in the moment pass (pair groups spread over blockIdx.z)
    const int* sepdone;     // (B, H, P) or NULL: 1 = the pair's moments were written by sep_grad_moments_kernel (skip it here)
    int share_cu;           // 1: launched as two 512-thread workgroups per CU (pair_moments_kernel, D <= 3)
    int mean_done;          // 1: msum was written by mean_moments_kernel (grad_stream_kernel.h): the streaming pass skips its mean part
    unsigned magic_N, magic_wpp;
    // one candidate evaluated for a host-side optimiser (gpmpc_objective_grad_host): the sweep copies host_n results from host_src
    // (device) to host_out (pinned host memory) when it is done and then stores host_flag_value at host_flag (host_n = 0: nothing)
    double* host_out; const double* host_src; int host_n;
    unsigned long long* host_flag; unsigned long long host_flag_value;
};

__host__ __device__ inline int tri_index(int d, int e, int DP) { return d * DP - (d * (d - 1)) / 2 + (e - d); }   // d <= e

// Mean-part moments per output a (sums over the points, weights lb_ai):
//   [0] 1 | [1, 1+D) nu_d | tri(D) nu_d1 nu_d2 (d1 <= d2) | D x tri(D) nu_k nu_d1 nu_d2 | NX nu_x | D x NX nu_k nu_x
__host__ __device__ inline int tri_count(int D) { return D * (D + 1) / 2; }
__host__ __device__ inline int mean_moment_count(int D, int NX) { return 1 + D + tri_count(D) + D * tri_count(D) + NX + D * NX; }
__host__ __device__ inline int sym_index(int i, int j, int D) { return i <= j ? tri_index(i, j, D) : tri_index(j, i, D); }
__device__ inline void decode_tri(int k, int D, int& d1, int& d2) {
    d1 = 0;
    while (k >= D - d1) { k -= D - d1; ++d1; }
    d2 = d1 + k;
}
// factors of component `comp`: indices < D are state dims, D + x the extra input dims; -1 = no factor
__device__ inline void decode_mean_moment(int comp, int D, int NX, int& i1, int& i2, int& i3) {
    const int T2 = tri_count(D);
    i1 = i2 = i3 = -1;
    if (comp == 0) return;
    comp -= 1;
    if (comp < D) { i1 = comp; return; }
    comp -= D;
    if (comp < T2) { decode_tri(comp, D, i1, i2); return; }
    comp -= T2;
    if (comp < D * T2) { i1 = comp / T2; decode_tri(comp - i1 * T2, D, i2, i3); return; }
    comp -= D * T2;
    if (comp < NX) { i1 = D + comp; return; }
    comp -= NX;
    i1 = comp / NX;
    i2 = D + (comp - i1 * NX);
}

// Row record of the gradient's moment pass: ka'_i | ra_i | g_i (DP) | u_i (DP) | nu_ie / l_ae^2 (NXP), padded to an EVEN number of
// doubles so that every record is 16-byte aligned and the compiler can read it as ds_read_b128 (4 LDS cycles per 16 bytes; an
// odd stride leaves it pairs of 8-byte reads, ds_read2_b64, at 8) -- the pass is within 25 % of the LDS return rate.
__host__ __device__ constexpr int grad_row_stride(int DP, int NXP) { return (2 + 2 * DP + NXP + 1) & ~1; }

struct MomLayout {
    int c_ils2, c_xr, c_logvar, c_tab, m, Sig, aug, ints, nu, xe, lb, kb, rows, part, total;
};

__host__ __device__ inline MomLayout make_mom_layout(int N, int D, int E, int G, int RS, int NR, int wpp, int NSP) {
    MomLayout L;
    const int P = D * (D + 1) / 2;
    int o = 0;
    L.c_ils2 = o;   o += rnd2(D * E);
    L.c_xr = o;     o += rnd2(2 * E);
    L.c_logvar = o; o += rnd2(D);
    L.c_tab = o;    o += 64;
    L.m = o;        o += rnd2(E);
    L.Sig = o;      o += rnd2(D * D);
    L.aug = o;      o += (D + G) * 2 * D * D;
    L.ints = o;     o += rnd2((3 * P + G + 4 + ((N + 3) / 4 + 2) + 1) / 2);      // pa[P], pb[P], K[G], counter, tri[RC + 1], pq[P], pn
    L.nu = o;       o += rnd2(D * N);
    L.xe = o;       o += rnd2((E - D) * N);
    L.lb = o;       o += rnd2(D * N);
    L.kb = o;       o += rnd2(G * N);
    L.rows = o;     o += G * NR * RS;
    L.part = o;     o += rnd2(G * wpp * NSP);
    L.total = o;
    return L;
}

// ------------------------------------------------------------------------------------------
// NT = 512 with one column per lane at D <= 3: the same 128-VGPR code as the 1024-thread workgroup, TWO workgroups per CU (when
// their LDS fits twice) -- one's per-step set-up and reductions overlap the other's item loop.
template <int DP, int NXP, int NT, int NC>
__device__ __forceinline__ void pair_moments_body(const GradArgs& p, double* smem) {
    constexpr int NW = NT / kWave;
    constexpr int RS = grad_row_stride(DP, NXP);     // row record: ka'_i, beta_ai, g_i (DP), u_i (DP), nu_ie / l_ae^2 (NXP), pad
    constexpr int NH = DP * (DP + 1) / 2;
    constexpr int NSP = 1 + DP + NH + NXP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = blockIdx.x, c = blockIdx.y;
    const int N = p.N, D = p.D, A = p.A, E = p.E, H = p.H, G = p.G;
    const int NX = E - D;
    const int P = D * (D + 1) / 2;
    const int LD = 2 * D;
    const int wpp = p.wpp;
    const int NCU = (N + NC - 1) / NC;      // column units per row chunk (NC adjacent columns per lane)
    const int NR = p.RC * p.CH;
    const MomLayout L = make_mom_layout(N, D, E, G, RS, NR, wpp, NSP);
    double* c_ils2 = smem + L.c_ils2;
    double* c_logvar = smem + L.c_logvar;
    double* c_tab = smem + L.c_tab;
    double* s_m = smem + L.m;
    double* s_Sig = smem + L.Sig;
    double* s_aug = smem + L.aug;
    int* s_pa = reinterpret_cast<int*>(smem + L.ints);
    int* s_pb = s_pa + P;
    int* s_K = s_pb + P;                 // per pair of the group: Taylor degree of exp(g.w), 0 = direct exp
    int* s_counter = s_K + G;
    int* s_tri = s_counter + 1;          // diagonal pairs: column units of row chunks < r that can hold an element i <= j
    int* s_pq = s_tri + ((N + 3) / 4 + 2);      // pair index (a <= b enumeration) of every pair this kernel works on
    int* s_pn = s_pq + P;                       // their number: all P, or those the separable kernel left (grad_sep_kernel.h)
    double* c_xr = smem + L.c_xr;
    double* a_nu = smem + L.nu;
    double* a_xe = smem + L.xe;
    double* a_lb = smem + L.lb;
    double* a_kb = smem + L.kb;
    double* a_rows = smem + L.rows;
    double* s_part = smem + L.part;

    {
        // constants and the step's state: one element of every array per thread, all loads in flight together (see rollout_kernel.h;
        // E <= NT, D * D <= NT)
        const double* msrc = (tid < D) ? p.mu + ((size_t)c * (H + 1) + t) * D + tid
                                       : p.actions + ((size_t)c * H + t) * A + ((tid < D + A) ? tid - D : 0);
        const double* ssrc = p.Sig + ((size_t)c * (H + 1) + t) * D * D + (tid < D * D ? tid : 0);
        __builtin_amdgcn_sched_barrier(0);
        const double v0 = p.logvar[tid < D ? tid : 0], v1 = p.ils2[tid < D * E ? tid : 0], v2 = kExp2Tab[tid & 63];
        const double v3 = p.xrange[tid < 2 * E ? tid : 0], v4 = *msrc, v5 = *ssrc;
        __builtin_amdgcn_sched_barrier(0);
        if (tid < D) c_logvar[tid] = v0;
        if (tid < D * E) c_ils2[tid] = v1;
        if (tid < 64) c_tab[tid] = v2;
        if (tid < 2 * E) c_xr[tid] = v3;
        if (tid < E) s_m[tid] = (tid < D + A) ? v4 : p.time0 + (double)t;
        if (tid < D * D) s_Sig[tid] = v5;
        for (int i = tid + NT; i < D * E; i += NT) c_ils2[i] = p.ils2[i];
    }
    for (int i = tid; i < G * (NR - N) * RS; i += NT) {
        const int per = (NR - N) * RS;
        const int gq = i / per, k = i - gq * per;
        a_rows[((size_t)gq * NR + N) * RS + k] = 0.0;                      // zero padding rows
    }
    if (tid == 0) {
        int q = 0, n = 0;
        for (int a = 0; a < D; ++a)
            for (int b = a; b < D; ++b, ++q) {
                if (p.sepdone && p.sepdone[((size_t)c * H + t) * P + q]) continue;
                s_pa[n] = a; s_pb[n] = b; s_pq[n] = q; ++n;
            }
        *s_pn = n;
        // a column unit (NC adjacent columns) is useful for row chunk r of a diagonal pair if its last column >= r CH
        int run = 0;
        for (int r = 0; r <= p.RC; ++r) {
            s_tri[r] = run;
            const int first = (r * p.CH) / NC;
            run += (first < NCU) ? NCU - first : 0;
        }
    }
    __syncthreads();
#if defined(GPMPC_PROF_ON)
    long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long prof_last = __builtin_readcyclecounter();
#define GPMPC_GTRACE(id) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) { long long now_ = __builtin_readcyclecounter(); \
    prof_acc[id] += now_ - prof_last; prof_last = now_; } } while (0)
#else
#define GPMPC_GTRACE(id) do {} while (0)
#endif

    // Z = R^-1 Sigma of a pair (gp_model.py:156-163) into its augmented block, and the Taylor degree of exp(g . w) from the bound
    // |g_i . w_j| <= cmax over the data range of the memory points (as in the forward kernel); one thread per pair
    auto solve_pair = [&](int gq, int q0) {
        const int a = s_pa[q0 + gq], b = s_pb[q0 + gq];
        double* aug = s_aug + (D + gq) * (D * LD);
        double cmax = 0.0;
        if constexpr (DP <= 4) {
            // the data range's bounds first (their LDS reads travel during the solve), Z stays in registers for the bound
            double ur[DP], wr[DP];
#pragma unroll
            for (int d = 0; d < DP; ++d) {
                const int dc = d < D ? d : 0;
                const double rg = fmax(fabs(c_xr[dc] - s_m[dc]), fabs(c_xr[E + dc] - s_m[dc]));
                ur[d] = (d < D) ? rg * c_ils2[a * E + dc] : 0.0;
                wr[d] = (d < D) ? rg * c_ils2[b * E + dc] : 0.0;
            }
            double m[DP][2 * DP];
#pragma unroll
            for (int i = 0; i < DP; ++i)
#pragma unroll
                for (int j = 0; j < DP; ++j) {
                    const bool in = (i < D && j < D);
                    const double sg = in ? s_Sig[i * D + j] : 0.0;
                    m[i][j] = sg * (in ? c_ils2[a * E + j] + c_ils2[b * E + j] : 0.0) + (i == j ? 1.0 : 0.0);     // R (:156-159)
                    m[i][DP + j] = sg;
                }
            (void)small_solve<DP>(m);                                                       // Z = R^-1 Sigma
#pragma unroll
            for (int i = 0; i < DP; ++i)
#pragma unroll
                for (int j = 0; j < DP; ++j)
                    if (i < D && j < D) {
                        aug[i * LD + D + j] = m[i][DP + j];
                        cmax = fma(fabs(m[i][DP + j]) * ur[i], wr[j], cmax);
                    }
        } else {
            for (int i = 0; i < D; ++i)
                for (int j = 0; j < D; ++j) {
                    const double dab = c_ils2[a * E + j] + c_ils2[b * E + j];
                    aug[i * LD + j] = s_Sig[i * D + j] * dab + (i == j ? 1.0 : 0.0);
                    aug[i * LD + D + j] = s_Sig[i * D + j];
                }
            (void)gauss_solve(aug, D, D, LD);
            for (int i = 0; i < D; ++i) {
                const double ui = fmax(fabs(c_xr[i] - s_m[i]), fabs(c_xr[E + i] - s_m[i])) * c_ils2[a * E + i];
                for (int j = 0; j < D; ++j) {
                    const double wj = fmax(fabs(c_xr[j] - s_m[j]), fabs(c_xr[E + j] - s_m[j])) * c_ils2[b * E + j];
                    cmax = fma(fabs(aug[i * LD + D + j]) * ui, wj, cmax);
                }
            }
        }
        int K = 0;
        if (p.force_path != 1 && cmax <= kTaylorMaxArg[kMaxTaylor]) {
            K = 1;
            for (int k = 1; k < kMaxTaylor; ++k) K += (cmax > kTaylorMaxArg[k]) ? 1 : 0;
        }
        s_K[gq] = K;
    };

    // Set-up.  The mean part (lb_ai and the third-order moments of nu under it) is formed here only when mean_moments_kernel has
    // not done it (p.mean_done): 162 wave reductions over LDS-resident points at config 2 -- a quarter of this kernel's time per
    // (candidate, step), bound by the LDS reads of its products (profiles/r04j_moment_phases.txt).  The small solves of the FIRST
    // pair group run on wavefront 1 beside the loads and the A_a solves of wavefront 0 instead of as a serial phase of their own.
    const bool mean_here = !p.mean_done;
    const int Pn = *s_pn;
    const int q_first = (int)blockIdx.z * G;
    for (int i = tid; i < D * N; i += NT) a_nu[i] = p.Xt[i] - s_m[i / N];
    for (int i = tid; i < NX * N; i += NT) a_xe[i] = p.Xt[(size_t)D * N + i] - s_m[D + i / N];
    if (mean_here && tid < D) {
        const int a = tid;
        double* aug = s_aug + a * (D * LD);
        if constexpr (DP <= 4) {
            double m[DP][2 * DP];
#pragma unroll
            for (int i = 0; i < DP; ++i)
#pragma unroll
                for (int j = 0; j < DP; ++j) {
                    const bool in = (i < D && j < D);
                    m[i][j] = (in ? s_Sig[i * D + j] : 0.0) + (i == j ? (i < D ? 1.0 / c_ils2[a * E + i] : 1.0) : 0.0);
                    m[i][DP + j] = (i == j) ? 1.0 : 0.0;
                }
            (void)small_solve<DP>(m);
#pragma unroll
            for (int i = 0; i < DP; ++i)
#pragma unroll
                for (int j = 0; j < DP; ++j)
                    if (i < D && j < D) aug[i * LD + D + j] = m[i][DP + j];
        } else {
            for (int i = 0; i < D; ++i)
                for (int j = 0; j < D; ++j) {
                    aug[i * LD + j] = s_Sig[i * D + j] + (i == j ? 1.0 / c_ils2[a * E + i] : 0.0);
                    aug[i * LD + D + j] = (i == j ? 1.0 : 0.0);
                }
            (void)gauss_solve(aug, D, D, LD);
        }
    }
    if (tid >= 64 && tid < 64 + G && q_first + (tid - 64) < Pn) solve_pair(tid - 64, q_first);
    if (tid == NT - 1) *s_counter = 0;
    __syncthreads();

    if (mean_here) {
        // mean part: lb_ai (gp_model.py:148) and its first moments
        for (int it = tid; it < D * N; it += NT) {
            const int a = it / N, pt = it - a * N;
            const double* Ai = s_aug + a * (D * LD) + D;
            double q = 0.0;
            for (int i = 0; i < D; ++i) {
                double r = 0.0;
                for (int j = 0; j < D; ++j) r = fma(Ai[i * LD + j], a_nu[j * N + pt], r);
                q = fma(a_nu[i * N + pt], r, q);
            }
            for (int x = 0; x < NX; ++x) {
                const double v = a_xe[x * N + pt];
                q = fma(v * v, c_ils2[a * E + D + x], q);
            }
            a_lb[it] = exp(-0.5 * q) * p.beta[it];
        }
        __syncthreads();
        GPMPC_GTRACE(0);
        // moments of nu under the weights lb_ai up to third order (what the reverse sweep needs of the mean part)
        const int NM = mean_moment_count(D, NX);
        for (int task = wave; task < ((blockIdx.z == 0) ? D * NM : 0); task += NW) {
            const int a = task / NM, comp = task - a * NM;
            int i1, i2, i3;
            decode_mean_moment(comp, D, NX, i1, i2, i3);
            const double* f1 = i1 < 0 ? nullptr : (i1 < D ? a_nu + i1 * N : a_xe + (i1 - D) * N);
            const double* f2 = i2 < 0 ? nullptr : (i2 < D ? a_nu + i2 * N : a_xe + (i2 - D) * N);
            const double* f3 = i3 < 0 ? nullptr : (i3 < D ? a_nu + i3 * N : a_xe + (i3 - D) * N);
            double v = 0.0;
            if (!f1) { for (int pt = lane; pt < N; pt += 64) v += a_lb[a * N + pt]; }
            else if (!f2) { for (int pt = lane; pt < N; pt += 64) v = fma(a_lb[a * N + pt], f1[pt], v); }
            else if (!f3) { for (int pt = lane; pt < N; pt += 64) v = fma(a_lb[a * N + pt] * f1[pt], f2[pt], v); }
            else { for (int pt = lane; pt < N; pt += 64) v = fma(a_lb[a * N + pt] * f1[pt] * f2[pt], f3[pt], v); }
            v = wave_sum(v);
            if (lane == 0) p.msum[(((size_t)c * H + t) * D + a) * NM + comp] = v;
        }
    }

    // small batches: the pair groups of one (candidate, step) are spread over gridDim.z workgroups (each repeats the
    // per-point set-up; the mean moments are written by z = 0)
    for (int q0 = q_first; q0 < Pn; q0 += G * (int)gridDim.z) {
        const int Gc = (Pn - q0 < G) ? (Pn - q0) : G;
        if (q0 != q_first) {                       // later groups: their small solves as a phase of their own
            if (tid < Gc) solve_pair(tid, q0);
            if (tid == NT - 1) *s_counter = 0;
            __syncthreads();
        }
        GPMPC_GTRACE(2);


        for (int it = tid; it < Gc * N; it += NT) {
            const int gq = it / N, pt = it - gq * N;
            const int a = s_pa[q0 + gq], b = s_pb[q0 + gq];
            const double* Z = s_aug + (D + gq) * (D * LD) + D;
            double u[DP], w[DP], g[DP];
            double ksa = 0.0, ksb = 0.0;
#pragma unroll
            for (int d = 0; d < DP; ++d) {
                const double nu = (d < D) ? a_nu[d * N + pt] : 0.0;
                u[d] = (d < D) ? nu * c_ils2[a * E + d] : 0.0;
                w[d] = (d < D) ? nu * c_ils2[b * E + d] : 0.0;
                ksa = fma(nu, u[d], ksa);
                ksb = fma(nu, w[d], ksb);
                g[d] = 0.0;
            }
            double* rec = a_rows + ((size_t)gq * NR + pt) * RS;
#pragma unroll
            for (int x = 0; x < NXP; ++x) {
                const double v = (x < NX) ? a_xe[x * N + pt] : 0.0;
                const double ia = (x < NX) ? c_ils2[a * E + D + x] : 0.0;
                const double ib = (x < NX) ? c_ils2[b * E + D + x] : 0.0;
                ksa = fma(v * v, ia, ksa);
                ksb = fma(v * v, ib, ksb);
                rec[2 + 2 * DP + x] = v * ia;
            }
            double qa = 0.0, qb = 0.0;
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
                            g[j] = fma(z, u[i], g[j]);
                        }
                    qa = fma(u[i], zu, qa);
                    qb = fma(w[i], zw, qb);
                }
            }
            const double ka = c_logvar[a] - 0.5 * ksa + 0.5 * qa;
            const double kb = c_logvar[b] - 0.5 * ksb + 0.5 * qb;
            const double ba = p.beta[a * N + pt];
            if (s_K[gq] > 0) {                         // Taylor form: exp(ka'), exp(kb') are per-point factors
                const double ea = exp(ka);
                rec[0] = ea;
                rec[1] = ea * ba;
                a_kb[gq * N + pt] = (a == b) ? 2.0 * ea : exp(kb) * p.beta[b * N + pt];
            } else {
                rec[0] = ka;
                rec[1] = ba;
                a_kb[gq * N + pt] = kb;
            }
#pragma unroll
            for (int d = 0; d < DP; ++d) { rec[2 + d] = g[d]; rec[2 + DP + d] = u[d]; }
        }
        __syncthreads();
        GPMPC_GTRACE(3);

        const int total = Gc * wpp;
        auto pull_item = [&]() -> int {
            int pulled = 0;
            if (lane == 0) pulled = __hip_atomic_fetch_add(s_counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return __builtin_amdgcn_readfirstlane(pulled);
        };
        for (int wi = pull_item(); wi < total; wi = pull_item()) {
            const int gq = p.magic_wpp ? (int)__umulhi((unsigned)wi, p.magic_wpp) : wi;
            const int slot = wi - gq * wpp;
            const int a = __builtin_amdgcn_readfirstlane(s_pa[q0 + gq]);
            const int b = __builtin_amdgcn_readfirstlane(s_pb[q0 + gq]);
            const bool diag = (a == b);
            if (diag && slot * 64 >= s_tri[p.RC]) {        // a slot beyond the triangle's tiles (wpp counts those of a full matrix)
                for (int k = lane; k < NSP; k += 64) s_part[(size_t)wi * NSP + k] = 0.0;
                continue;
            }
            const int flat = slot * 64 + lane;
            bool valid;
            int r, jc;
            if (diag) {
                // only the (row chunk, column unit) tiles that contain an element i <= j (as in the forward kernel)
                valid = flat < s_tri[p.RC];
                int lo = 0, hi = p.RC;
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (s_tri[mid] <= flat) lo = mid; else hi = mid;
                }
                r = valid ? lo : 0;
                jc = valid ? (r * p.CH) / NC + (flat - s_tri[r]) : 0;
            } else {
                valid = flat < p.RC * NCU;
                r = valid ? (p.magic_N ? (int)__umulhi((unsigned)flat, p.magic_N) : flat) : 0;       // flat / NCU
                jc = valid ? flat - r * NCU : 0;
            }
            // the lane's NC adjacent columns j .. j + NC - 1 (two columns per lane halve the LDS broadcast traffic per
            // element, which bounds the one-column loop: 72 bytes of row record per lane and row)
            const int j = NC * jc;
            bool vcol[NC];
#pragma unroll
            for (int q = 0; q < NC; ++q) vcol[q] = valid && (j + q < N);
            const int jl = (NC == 2 && vcol[NC - 1]) ? j + 1 : j;          // last column of the lane
            const int i0 = r * p.CH;
            int i1 = i0 + p.CH;
            if (i1 > N) i1 = N;
            if (diag && i1 > jl + 1) i1 = jl + 1;
            const int len = (valid && i1 > i0) ? (i1 - i0) : 0;
            const int nrows = (wave_max_i32(len) + 3) & ~3;
            double cs[NC], h[NC][DP + NXP], hh[NC][NH], w[NC][DP], kbj[NC];
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                cs[q] = 0.0;
#pragma unroll
                for (int d = 0; d < DP + NXP; ++d) h[q][d] = 0.0;
#pragma unroll
                for (int k = 0; k < NH; ++k) hh[q][k] = 0.0;
#pragma unroll
                for (int d = 0; d < DP; ++d) w[q][d] = (d < D && vcol[q]) ? a_nu[d * N + j + q] * c_ils2[b * E + d] : 0.0;
                kbj[q] = vcol[q] ? a_kb[gq * N + j + q] : 0.0;
            }
            const int K = __builtin_amdgcn_readfirstlane(s_K[gq]);
            if (nrows > 0) {
                const double* rec = a_rows + ((size_t)gq * NR + i0) * RS;
                const double* Tp = p.Tm + ((size_t)a * (N + kTPad) + i0) * N + j;
                auto accumulate = [&](int q, double e, const double (&rr)[RS]) {
                    cs[q] += e;
                    int k = 0;
#pragma unroll
                    for (int d = 0; d < DP; ++d) {
                        const double td = e * rr[2 + DP + d];
                        h[q][d] += td;
#pragma unroll
                        for (int d2 = d; d2 < DP; ++d2) { hh[q][k] = fma(td, rr[2 + DP + d2], hh[q][k]); ++k; }
                    }
#pragma unroll
                    for (int x = 0; x < NXP; ++x) h[q][DP + x] = fma(e, rr[2 + 2 * DP + x], h[q][DP + x]);
                };
                // two rows per trip; the next trip's T values are in flight during this trip's math (reading past the
                // last trip only touches the zero padding rows of T)
                auto run = [&](auto kc) {
                    constexpr int KK = decltype(kc)::value;
                    double tn[2][NC];
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int q = 0; q < NC; ++q) tn[u][q] = diag ? Tp[(size_t)u * N + q] : 1.0;
                    for (int it = 0; it < nrows; it += 2) {
                        double e[2][NC];
                        double rv[2][RS];                      // the two row records, read as ds_read_b128 (grad_row_stride)
#pragma unroll
                        for (int u = 0; u < 2; ++u) lds_load_pairs<2 + 2 * DP + NXP>(rec + u * RS, rv[u]);
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const double(&rr)[RS] = rv[u];
#pragma unroll
                            for (int q = 0; q < NC; ++q) {
                                const double tv = tn[u][q];
                                tn[u][q] = diag ? Tp[(size_t)(2 + u) * N + q] : 1.0;
                                if constexpr (KK > 0) {
                                    double cc = rr[2] * w[q][0];
#pragma unroll
                                    for (int d = 1; d < DP; ++d) cc = fma(rr[2 + d], w[q][d], cc);
                                    e[u][q] = taylor_exp<KK>(cc) * (diag ? rr[0] * tv : rr[1]);
                                } else {
                                    double arg = rr[0] + kbj[q];
#pragma unroll
                                    for (int d = 0; d < DP; ++d) arg = fma(rr[2 + d], w[q][d], arg);
                                    e[u][q] = fast_exp(arg, c_tab) * (diag ? tv : rr[1]);
                                }
                            }
                        }
#pragma unroll
                        for (int u = 0; u < 2; ++u)
#pragma unroll
                            for (int q = 0; q < NC; ++q) accumulate(q, e[u][q], rv[u]);
                        rec += 2 * RS;
                        Tp += (size_t)2 * N;
                    }
                };
                if (K == 0) run(std::integral_constant<int, 0>{});
                else if (K <= 2) run(std::integral_constant<int, 2>{});
                else if (K == 3) run(std::integral_constant<int, 3>{});
                else if (K == 4) run(std::integral_constant<int, 4>{});
                else if (K == 5) run(std::integral_constant<int, 5>{});
                else if (K == 6) run(std::integral_constant<int, 6>{});
                else if (K <= 8) run(std::integral_constant<int, 8>{});
                else if (K <= 10) run(std::integral_constant<int, 10>{});
                else if (K <= 12) run(std::integral_constant<int, 12>{});
                else run(std::integral_constant<int, 14>{});
            }
            // column factor; for a diagonal pair only i <= j was visited with a halved diagonal of T and every
            // moment is symmetric under i <-> j, so the factor is 2 (Taylor form: already in kbj)
            double colf[NC], xb[NC][NXP];
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                colf[q] = vcol[q] ? (K > 0 ? kbj[q] : (diag ? 2.0 : p.beta[b * N + j + q])) : 0.0;
#pragma unroll
                for (int x = 0; x < NXP; ++x) xb[q][x] = (x < NX && vcol[q]) ? a_xe[x * N + j + q] * c_ils2[b * E + D + x] : 0.0;
            }
            // f(column) * colf summed over the lane's columns, then the NSP lane values over the wavefront: sixteen at a time on the
            // halving exchange network (wave_reduce16: 57 instructions per 16 values, lane l ends with the total of value l >> 2)
            // instead of one 6-step butterfly per value -- the fold was a fifth of a 64-row item
            double* out = s_part + (size_t)wi * NSP;
            double fv[NSP];
            auto fold = [&](int slot, auto f) {
                double v = 0.0;
#pragma unroll
                for (int q = 0; q < NC; ++q) v = fma(f(q), colf[q], v);
                fv[slot] = v;
            };
            fold(0, [&](int q) { return cs[q]; });
            int k = 0;
#pragma unroll
            for (int d = 0; d < DP; ++d) {
                fold(1 + d, [&](int q) { return h[q][d] + cs[q] * w[q][d]; });
#pragma unroll
                for (int d2 = d; d2 < DP; ++d2) {
                    fold(1 + DP + k, [&](int q) { return hh[q][k] + cs[q] * w[q][d] * w[q][d2] + h[q][d] * w[q][d2] + w[q][d] * h[q][d2]; });
                    ++k;
                }
            }
#pragma unroll
            for (int x = 0; x < NXP; ++x) fold(1 + DP + NH + x, [&](int q) { return h[q][DP + x] + cs[q] * xb[q][x]; });
            constexpr int NFG = NSP / 16, NFR = NSP % 16;         // full groups of 16; the rest: one more group, or single sums when it is short
            if constexpr (NSP <= 8) {
                double grp[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) grp[e] = (e < NSP) ? fv[e] : 0.0;
                const double tot = wave_reduce8(grp);
                if ((lane & 3) == 0 && lane < 32 && (lane >> 2) < NSP) out[lane >> 2] = tot;
            } else
#pragma unroll
            for (int g = 0; g < NFG + ((NFR > 2) ? 1 : 0); ++g) {
                double grp[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) grp[e] = (16 * g + e < NSP) ? fv[16 * g + e] : 0.0;
                const double tot = wave_reduce16(grp);
                const int slot = 16 * g + (lane >> 2);
                if ((lane & 3) == 0 && slot < NSP) out[slot] = tot;
            }
            if constexpr (NSP > 8 && NFR > 0 && NFR <= 2) {
#pragma unroll
                for (int e = 0; e < NFR; ++e) {
                    const double tot = wave_sum(fv[16 * NFG + e]);
                    if (lane == 0) out[16 * NFG + e] = tot;
                }
            }
        }
        GPMPC_GTRACE(4);
        __syncthreads();
        GPMPC_GTRACE(5);

        for (int task = wave; task < Gc * NSP; task += NW) {
            const int gq = task / NSP, k = task - gq * NSP;
            double v = 0.0;
            for (int s = lane; s < wpp; s += 64) v += s_part[((size_t)gq * wpp + s) * NSP + k];
            v = wave_sum(v);
            if (lane == 0) p.mom[((((size_t)c * H + t) * P) + s_pq[q0 + gq]) * NSP + k] = v;
        }
        __syncthreads();
        GPMPC_GTRACE(6);
    }
#if defined(GPMPC_PROF_ON)
    if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0)
        printf("PROF moments cycles: nu/lb %lld | mean moments + Z %lld | records %lld | items(wave0) %lld | wait %lld | fold %lld\n",
               prof_acc[0], prof_acc[2], prof_acc[3], prof_acc[4], prof_acc[5], prof_acc[6]);
#endif
}

template <int DP, int NXP, int NT, int NC>
__global__ __launch_bounds__(NT, (NT == 512 && NC == 1 && DP <= 3) ? 4 : 1) void pair_moments_kernel(const GradArgs p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    pair_moments_body<DP, NXP, NT, NC>(p, smem);
}

// ------------------------------------------------------------------------------------------
constexpr int kSweepAug = 16;     // LDS augmented blocks of the sweep (D > 4)
// registers per lane that hold the next step's moments: max(P * NSP, D * NM) / NT for the largest shapes of a padded D
// (NX <= 6): DP 2: 3*14 | 2*24 -> 1 (64 lanes);  3: 6*19 | 3*54 -> 3;  4: 10*25 | 4*94 -> 6;  6: 21*40 | 6*223 -> 6 (256 lanes);  8: 36*57 | 8*423 -> 14
// (four sweep wavefronts at DP <= 4: 256 lanes -> 1, 1, 2)
template <int DP, int NT>
constexpr int kSweepPrefetch = (DP <= 4 && NT >= 256) ? (DP == 4 ? 2 : 1) : (DP == 2) ? 1 : (DP == 3) ? 3 : (DP == 4) ? 6 : (DP == 6) ? 6 : 14;

struct SweepLayout {
    int c_ils2, c_var, cost, gmu, gSig, gu, ctmp, mubar, Sigbar, Sacc, mbar, m, Sig, ms, mom, Ai, cc, M, y, V, Sb, Vb, Mb, cb, s0b,
        s1b, Gs, Aib, Ab, mba, Ri, Z, rdet, RZ, Kq, mq, aug, pre, tmu, tSig, tact, tcv, total;
};

__host__ __device__ inline int sweep_pre_words(int D) {        // per step: Ai, c, Ri, Z, rdet, y, V, M
    const int P = D * (D + 1) / 2, DD = D * D;
    return D * DD + D + 2 * P * DD + P + 2 * DD + D;
}

__host__ __device__ inline SweepLayout make_sweep_layout(int D, int A, int E, int H, int NSP, int nwaves, int naug, int pre_steps) {
    SweepLayout L;
    const int P = D * (D + 1) / 2, DD = D * D, n = D + A, NX = E - D;
    int o = 0;
    auto take = [&](int& f, int sz) { f = o; o += rnd2(sz); };
    take(L.c_ils2, D * E); take(L.c_var, D); take(L.cost, n + n * n + DD + 2 * D);
    take(L.gmu, (H + 1) * D); take(L.gSig, (H + 1) * DD); take(L.gu, H * A); take(L.ctmp, nwaves * (2 * n * n + 3 * n));
    take(L.mubar, D); take(L.Sigbar, DD); take(L.Sacc, DD); take(L.mbar, E);
    take(L.m, E); take(L.Sig, DD); take(L.ms, D * mean_moment_count(D, NX)); take(L.mom, P * NSP);
    take(L.Ai, D * DD); take(L.cc, D); take(L.M, D);
    take(L.y, DD); take(L.V, DD); take(L.Sb, DD); take(L.Vb, DD); take(L.Mb, D); take(L.cb, D); take(L.s0b, D); take(L.s1b, DD);
    take(L.Gs, D * (D + DD + NX)); take(L.Aib, D * DD); take(L.Ab, D * DD); take(L.mba, D * E);
    take(L.Ri, P * DD); take(L.Z, P * DD); take(L.rdet, P); take(L.RZ, P * DD); take(L.Kq, D <= 4 ? P * DD : 0); take(L.mq, P * E);
    take(L.aug, naug * D * 2 * D);
    take(L.pre, pre_steps * sweep_pre_words(D));      // state-independent small algebra of every step, computed up front
    // the candidate's stored trajectory, actions and cost variances, staged once: the prologue's cost adjoints read them in
    // dependent inner loops (from global memory they were ~30 k of its ~60 k cycles at config 2)
    take(L.tmu, (H + 1) * D); take(L.tSig, (H + 1) * DD); take(L.tact, H * A); take(L.tcv, H + 1);
    L.total = o;
    return L;
}

// Partials of the stage cost (setpoint_distance_reward_mapper.py:36-66; terminal :135-141) wrt (mu, Sigma, u),
// already weighted by dJ/dcm = 1/(H+1), dJ/dcv = -kappa / (2 sqrt(cv) (H+1)) (gp_mpc_controller.py:270-276).
// One wavefront per time step; `tmp` is that wave's scratch (2 n^2 + 3 n doubles).
__device__ inline void cost_adjoint_wave(int lane, int D, int A, bool terminal, const double* mu, const double* Sg, const double* act,
                                         const double* target, const double* Wm, const double* smin, const double* smax,
                                         bool use_constraints, double wm, double wv, double* tmp, double* gmu, double* gSig,
                                         double* gu) {
    const int n = terminal ? D : D + A;
    double* WS = tmp;            // W Sa  (n x n, Sa = state block)
    double* Gm = WS + n * n;     // W Sa W
    double* err = Gm + n * n;
    double* We = err + n;
    double* WTe = We + n;
    for (int i = lane; i < n; i += 64) err[i] = (i < D ? mu[i] : act[i - D]) - target[i];
    for (int idx = lane; idx < n * n; idx += 64) {
        const int i = idx / n, j = idx - i * n;
        double v = 0.0;
        if (j < D)
            for (int k = 0; k < D; ++k) v = fma(Wm[i * n + k], Sg[k * D + j], v);
        WS[idx] = v;
    }
    wave_lds_sync();
    for (int idx = lane; idx < n * n; idx += 64) {
        const int i = idx / n, j = idx - i * n;
        double v = 0.0;
        for (int k = 0; k < D; ++k) v = fma(WS[i * n + k], Wm[k * n + j], v);
        Gm[idx] = v;
    }
    for (int i = lane; i < n; i += 64) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < n; ++k) { a = fma(Wm[i * n + k], err[k], a); b = fma(Wm[k * n + i], err[k], b); }
        We[i] = a; WTe[i] = b;
    }
    wave_lds_sync();
    for (int idx = lane; idx < D * D; idx += 64) {
        const int i = idx / D, j = idx - i * D;
        // d cm / d Sa = W^T;  d cv / d Sa = 4 (W Sa W)^T + 4 (W^T e)(W e)^T
        double v = wm * Wm[j * n + i] + wv * 4.0 * (Gm[j * n + i] + WTe[i] * We[j]);
        if (use_constraints && !terminal && i == j) {
            const double sq = Sg[i * D + i];                 // the reference passes the variance as sigma (:60-64)
            const double zmin = (smin[i] - mu[i]) / sq, zmax = (smax[i] - mu[i]) / sq;
            const double pmin = exp(-0.5 * zmin * zmin) * 0.3989422804014327, pmax = exp(-0.5 * zmax * zmax) * 0.3989422804014327;
            v += wm * (-pmin * zmin + pmax * zmax) / sq;
        }
        gSig[idx] = v;
    }
    for (int i = lane; i < n; i += 64) {
        double v = 0.0;
        for (int k = 0; k < n; ++k) v = fma(Gm[i * n + k] + Gm[k * n + i], err[k], v);
        v = wm * (We[i] + WTe[i]) + wv * 4.0 * v;
        if (i < D) {
            if (use_constraints && !terminal) {
                const double sq = Sg[i * D + i];
                const double zmin = (smin[i] - mu[i]) / sq, zmax = (smax[i] - mu[i]) / sq;
                v += wm * (-exp(-0.5 * zmin * zmin) + exp(-0.5 * zmax * zmax)) * 0.3989422804014327 / sq;
            }
            gmu[i] = v;
        } else {
            gu[i - D] = v;
        }
    }
}

// One workgroup per candidate.  D <= 4: the reverse sweep itself runs on ONE wavefront (every hand-off is then a wave-level LDS
// sync), but the prologue -- the cost adjoints of the H + 1 time steps and the state-independent small algebra of every step,
// all independent of each other -- is spread over all NT / 64 wavefronts of the launch (round 6: on one wavefront it was half of the
// kernel at config 2, 58 of 123 us at B = 1); the other wavefronts leave before the sweep.
// DX = exact state dimension at compile time (0: runtime p.D <= DP): the index arithmetic of this latency-bound kernel
// (divisions by D and D*D, pair decoding) then folds into constants.
template <int DP, int NT, int DX>
__global__ __launch_bounds__(NT) void adjoint_sweep_kernel(const GradArgs p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int NW = NT / kWave;
    // threads of the reverse sweep proper.  D <= 4: four wavefronts when the launch has them -- the independent loops of a phase (e.g.
    // the mean part's s1_bar, the pairs' K_q and m_q) then run side by side on their own wavefronts instead of one after the other
    // on one (round 6; same arithmetic per element, results bit-identical) -- else the one
    constexpr int NL = (DP <= 4) ? (NT >= 256 ? 256 : 64) : NT;
    // first thread of the wavefront a loop is given to (0 when the sweep has a single wavefront)
    [[maybe_unused]] constexpr int W1 = (DP <= 4 && NL > 64) ? 64 : 0, W2 = (DP <= 4 && NL > 64) ? 128 : 0, W3 = (DP <= 4 && NL > 64) ? 192 : 0;
    constexpr int NPF = kSweepPrefetch<DP, NL>;
    // LDS-only hand-off: the fences name the local address space, so a sync does not wait for the global loads of the
    // next step that are in flight
    auto sync_all = [] {                             // prologue: all wavefronts of the launch
        if constexpr (NT == 64) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        }
    };
    auto sync = [] {
        if constexpr (NL == 64) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        }
    };
#if defined(GPMPC_PROF_ON)
    long long sprof[16] = {0};
    long long sprof_last = __builtin_readcyclecounter();
#define GPMPC_STRACE(id) do { if (threadIdx.x == 0 && blockIdx.x == 0) { long long now_ = __builtin_readcyclecounter(); \
    sprof[id] += now_ - sprof_last; sprof_last = now_; } } while (0)
#else
#define GPMPC_STRACE(id) do {} while (0)
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = blockIdx.x;
    const int D = (DX > 0) ? DX : p.D;
    const int A = p.A, E = p.E, H = p.H;
    const int NX = E - D, P = D * (D + 1) / 2, DD = D * D, n = D + A;
    const int NSP = p.NSP, NH = DP * (DP + 1) / 2;
    const int NG = D + DD + NX;
    const int T2 = tri_count(D), NM = mean_moment_count(D, NX);
    const int LD = 2 * D;
    const SweepLayout L = make_sweep_layout(D, A, E, H, NSP, NW, DP <= 4 ? 0 : kSweepAug, p.pre_steps);
    double* c_ils2 = smem + L.c_ils2; double* c_var = smem + L.c_var; double* c_cost = smem + L.cost;
    double* gmu = smem + L.gmu; double* gSig = smem + L.gSig; double* gu = smem + L.gu; double* ctmp = smem + L.ctmp;
    double* mubar = smem + L.mubar; double* Sigbar = smem + L.Sigbar; double* Sacc = smem + L.Sacc; double* mbar = smem + L.mbar;
    double* s_m = smem + L.m; double* s_Sig = smem + L.Sig; double* s_ms = smem + L.ms; double* s_mom = smem + L.mom;
    double* s_Ai = smem + L.Ai; double* s_cc = smem + L.cc; double* s_M = smem + L.M; double* s_y = smem + L.y; double* s_V = smem + L.V;
    double* s_pre = smem + L.pre;
    const bool pre = (DP <= 4) && p.pre_steps > 0;
    const int NQ = sweep_pre_words(D);
    double* s_Sb = smem + L.Sb; double* s_Vb = smem + L.Vb; double* s_Mb = smem + L.Mb; double* s_cb = smem + L.cb;
    double* s_s0b = smem + L.s0b; double* s_s1b = smem + L.s1b; double* s_Gs = smem + L.Gs; double* s_Aib = smem + L.Aib;
    double* s_Ab = smem + L.Ab; double* s_mba = smem + L.mba; double* s_Ri = smem + L.Ri; double* s_Z = smem + L.Z;
    double* s_rdet = smem + L.rdet; double* s_RZ = smem + L.RZ; double* s_mq = smem + L.mq;
    double* s_Kq = (DP <= 4) ? smem + L.Kq : s_RZ;          // K_q: own buffer, or in place over RZ
    [[maybe_unused]] double* s_aug = smem + L.aug;
    const double* target = c_cost;
    const double* Wst = c_cost + n;
    const double* WT = Wst + n * n;
    const double* smin = WT + DD;
    const double* smax = smin + D;
    const double* traj_mu = p.mu + (size_t)c * (H + 1) * D;
    const double* traj_Sig = p.Sig + (size_t)c * (H + 1) * DD;
    const double* act = p.actions + (size_t)c * H * A;
    const double* cvv = p.cv + (size_t)c * (H + 1);
    const double inv_n = 1.0 / (double)(H + 1);
    auto pair_of = [&](int q, int& a, int& b) { decode_tri(q, D, a, b); };

    double* s_tmu = smem + L.tmu; double* s_tSig = smem + L.tSig; double* s_tact = smem + L.tact; double* s_tcv = smem + L.tcv;
    {
        // constants and the stored trajectory: the first element of every array per thread with all loads in flight together (one
        // copy loop after the other is one global round trip after the other: seven of them, ~5 k cycles, round 6)
        const int nc = n + n * n + DD + 2 * D, n_mu = (H + 1) * D, n_Sig = (H + 1) * DD, n_act = H * A, n_cv = H + 1;
        __builtin_amdgcn_sched_barrier(0);
        const double v0 = p.ils2[tid < D * E ? tid : 0], v1 = p.var[tid < D ? tid : 0], v2 = p.cost[tid < nc ? tid : 0];
        const double v3 = traj_mu[tid < n_mu ? tid : 0], v4 = traj_Sig[tid < n_Sig ? tid : 0];
        const double v5 = act[tid < n_act ? tid : 0], v6 = cvv[tid < n_cv ? tid : 0];
        __builtin_amdgcn_sched_barrier(0);
        if (tid < D * E) c_ils2[tid] = v0;
        if (tid < D) c_var[tid] = v1;
        if (tid < nc) c_cost[tid] = v2;
        if (tid < n_mu) s_tmu[tid] = v3;
        if (tid < n_Sig) s_tSig[tid] = v4;
        if (tid < n_act) s_tact[tid] = v5;
        if (tid < n_cv) s_tcv[tid] = v6;
        for (int i = tid + NT; i < D * E; i += NT) c_ils2[i] = p.ils2[i];
        for (int i = tid + NT; i < nc; i += NT) c_cost[i] = p.cost[i];
        for (int i = tid + NT; i < n_mu; i += NT) s_tmu[i] = traj_mu[i];
        for (int i = tid + NT; i < n_Sig; i += NT) s_tSig[i] = traj_Sig[i];
        for (int i = tid + NT; i < n_act; i += NT) s_tact[i] = act[i];
        for (int i = tid + NT; i < n_cv; i += NT) s_tcv[i] = cvv[i];
    }
    sync_all();
    GPMPC_STRACE(10);
    // cost adjoints of every time step (independent of the sweep): one wavefront per step
    for (int t = wave; t <= H; t += NW) {
        const bool terminal = (t == H);
        const double wv = -p.kappa / (2.0 * sqrt(s_tcv[t])) * inv_n;
        cost_adjoint_wave(lane, D, A, terminal, s_tmu + t * D, s_tSig + t * DD, s_tact + (terminal ? 0 : t) * A, target,
                          terminal ? WT : Wst, smin, smax, p.use_constraints != 0, inv_n, wv, ctmp + wave * (2 * n * n + 3 * n),
                          gmu + t * D, gSig + t * DD, terminal ? ctmp + wave * (2 * n * n + 3 * n) : gu + t * A);
        wave_lds_sync();
    }
    sync_all();
    GPMPC_STRACE(11);
    for (int i = tid; i < D; i += NT) mubar[i] = gmu[H * D + i];
    for (int i = tid; i < DD; i += NT) {
        const int r = i / D, q = i - r * D;
        Sigbar[i] = 0.5 * (gSig[H * DD + i] + gSig[H * DD + q * D + r]);
    }
    sync_all();

    // ---- state-independent small algebra of every step, all steps in parallel (lanes over (step, problem)) -----------
    const int oCC = D * DD, oRI = oCC + D, oZ = oRI + P * DD, oRD = oZ + P * DD, oY = oRD + P, oV = oY + DD, oM = oV + DD;
    if constexpr (DP <= 4) {
        if (pre) {
            for (int idx = tid; idx < H * (D + P); idx += NT) {
                const int t = idx / (D + P), prob = idx - t * (D + P);
                const double* Sg = s_tSig + t * DD;
                double* base = s_pre + t * NQ;
                int a = prob, b = prob;
                if (prob >= D) pair_of(prob - D, a, b);
                double m[DP][2 * DP];
#pragma unroll
                for (int i = 0; i < DP; ++i)
#pragma unroll
                    for (int j = 0; j < DP; ++j) {
                        const bool in = (i < D && j < D);
                        const double sg = in ? Sg[i * D + j] : 0.0;
                        double v;
                        if (prob < D) v = sg + ((i == j) ? (i < D ? 1.0 / c_ils2[a * E + i] : 1.0) : 0.0);
                        else v = sg * (in ? c_ils2[a * E + j] + c_ils2[b * E + j] : 0.0) + (i == j ? 1.0 : 0.0);
                        m[i][j] = v;
                        m[i][DP + j] = (i == j) ? 1.0 : 0.0;
                    }
                const double det = small_solve<DP>(m);
                double* dst = (prob < D) ? base + a * DD : base + oRI + (prob - D) * DD;
#pragma unroll
                for (int i = 0; i < DP; ++i)
#pragma unroll
                    for (int j = 0; j < DP; ++j)
                        if (i < D && j < D) dst[i * D + j] = m[i][DP + j];
                if (prob < D) {
                    double prodil = 1.0;
                    for (int i = 0; i < D; ++i) prodil *= c_ils2[a * E + i];
                    base[oCC + a] = c_var[a] / sqrt(det * prodil);
                } else {
                    base[oRD + prob - D] = 1.0 / sqrt(det);
                }
            }
            sync_all();
            for (int idx = tid; idx < H * P * DD; idx += NT) {
                const int t = idx / (P * DD), i = idx - t * (P * DD);
                const int q = i / DD, r = (i - q * DD) / D, cc = i - q * DD - r * D;
                const double* base = s_pre + t * NQ;
                double v = 0.0;
                for (int k = 0; k < D; ++k) v = fma(base[oRI + q * DD + r * D + k], s_tSig[t * DD + k * D + cc], v);
                s_pre[t * NQ + oZ + i] = v;
            }
            for (int idx = tid; idx < H * DD; idx += NT) {
                const int t = idx / DD, i = idx - t * DD;
                const int a = i / D, k = i - a * D;
                const double* ms = p.msum + (((size_t)c * H + t) * D + a) * NM;
                double* base = s_pre + t * NQ;
                double v = 0.0;
                for (int j = 0; j < D; ++j) v = fma(base[a * DD + k * D + j], ms[1 + j], v);
                base[oY + a * D + k] = v;
                base[oV + k * D + a] = base[oCC + a] * v;
                if (k == 0) base[oM + a] = base[oCC + a] * ms[0];
            }
            sync_all();
        }
    }
    GPMPC_STRACE(12);
    // NPF values per lane cover P*NSP and D*NM (host checks the bound); E, D*D <= NL
    if constexpr (NL < NT) { if (tid >= NL) return; }          // the sweep proper runs on NL threads
    double pf_mom[NPF], pf_ms[NPF], pf_m = 0.0, pf_Sig = 0.0;
    auto fetch = [&](int t) {
        const double* mom = p.mom + ((size_t)c * H + t) * P * NSP;
        const double* ms = p.msum + ((size_t)c * H + t) * D * NM;
        // one load per value, NO branch between or around the loads (lanes past an array's end read its last element: never
        // stored): the memory counter is in-order, and after a join of divergent paths the compiler waits for EVERYTHING outstanding
        // before it touches a register a load may still write -- the next step's loads, issued a few instructions earlier, were
        // waited for right here (round 6: ~1.6 k of a step's 7.5 k cycles)
        const double* msrc = (tid < D) ? traj_mu + t * D + tid : act + t * A + (tid < D + A ? tid - D : 0);
        const double* ssrc = traj_Sig + t * DD + (tid < DD ? tid : 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int i = tid + k * NL;
            pf_mom[k] = mom[i < P * NSP ? i : P * NSP - 1];
            pf_ms[k] = ms[i < D * NM ? i : D * NM - 1];
        }
        pf_m = *msrc;                               // lanes >= D + A read a valid dummy; selected away when consumed
        pf_Sig = *ssrc;
        __builtin_amdgcn_sched_barrier(0);
    };
    fetch(H - 1);
    for (int t = H - 1; t >= 0; --t) {
        // ---- forward quantities of step t: already in registers (loaded one step ahead); the loads of step t-1 are
        //      issued here and stay in flight during this step's algebra --------------------------------------
        {
#pragma unroll
            for (int k = 0; k < NPF; ++k) {
                const int i = tid + k * NL;
                if (i < P * NSP) s_mom[i] = pf_mom[k];
                if (i < D * NM) s_ms[i] = pf_ms[k];
            }
            if (tid < E) s_m[tid] = (tid < D + A) ? pf_m : p.time0 + (double)t;
            if (tid < DD) s_Sig[tid] = pf_Sig;
            if (t > 0) fetch(t - 1);
            if (pre) {                       // this step's A^-1, c, R^-1, Z, 1/sqrt det R, y, V, M were computed up front
                double* base = s_pre + t * NQ;
                s_Ai = base; s_cc = base + oCC; s_Ri = base + oRI; s_Z = base + oZ; s_rdet = base + oRD;
                s_y = base + oY; s_V = base + oV; s_M = base + oM;
                for (int i = tid; i < DD; i += NL) {
                    const int r = i / D, q = i - r * D;
                    s_Sb[i] = 0.5 * (Sigbar[i] + Sigbar[q * D + r]);
                }
            }
        }
        sync();
        GPMPC_STRACE(0);
        if (!pre) {
        // ---- small solves: A_a^-1, c_a;  R_ab^-1, Z_ab, 1/sqrt det R_ab --------------------------------
        constexpr int PSTEP = (DP <= 4) ? NL : kSweepAug;        // LDS solves: kSweepAug threads, one augmented block each
        for (int prob = tid; prob < D + P && tid < PSTEP; prob += PSTEP) {
            int a = prob, b = prob;
            if (prob >= D) pair_of(prob - D, a, b);
            double det;
            double* dst = (prob < D) ? s_Ai + a * DD : s_Ri + (prob - D) * DD;
            if constexpr (DP <= 4) {
                double m[DP][2 * DP];
#pragma unroll
                for (int i = 0; i < DP; ++i)
#pragma unroll
                    for (int j = 0; j < DP; ++j) {
                        const bool in = (i < D && j < D);
                        const double sg = in ? s_Sig[i * D + j] : 0.0;
                        double v;
                        if (prob < D) v = sg + ((i == j) ? (i < D ? 1.0 / c_ils2[a * E + i] : 1.0) : 0.0);
                        else v = sg * (in ? c_ils2[a * E + j] + c_ils2[b * E + j] : 0.0) + (i == j ? 1.0 : 0.0);
                        m[i][j] = v;
                        m[i][DP + j] = (i == j) ? 1.0 : 0.0;
                    }
                det = small_solve<DP>(m);
#pragma unroll
                for (int i = 0; i < DP; ++i)
#pragma unroll
                    for (int j = 0; j < DP; ++j)
                        if (i < D && j < D) dst[i * D + j] = m[i][DP + j];
            } else {
                double* aug = s_aug + tid * (D * LD);
                for (int i = 0; i < D; ++i)
                    for (int j = 0; j < D; ++j) {
                        const double sg = s_Sig[i * D + j];
                        aug[i * LD + j] = (prob < D) ? sg + (i == j ? 1.0 / c_ils2[a * E + i] : 0.0)
                                                     : sg * (c_ils2[a * E + j] + c_ils2[b * E + j]) + (i == j ? 1.0 : 0.0);
                        aug[i * LD + D + j] = (i == j) ? 1.0 : 0.0;
                    }
                det = gauss_solve(aug, D, D, LD);
                for (int i = 0; i < D; ++i)
                    for (int j = 0; j < D; ++j) dst[i * D + j] = aug[i * LD + D + j];
            }
            if (prob < D) {
                double prodil = 1.0;
                for (int i = 0; i < D; ++i) prodil *= c_ils2[a * E + i];
                s_cc[a] = c_var[a] / sqrt(det * prodil);
            } else {
                s_rdet[prob - D] = 1.0 / sqrt(det);
            }
        }
        for (int i = tid; i < DD; i += NL) {
            const int r = i / D, q = i - r * D;
            s_Sb[i] = 0.5 * (Sigbar[i] + Sigbar[q * D + r]);
        }
        sync();
        GPMPC_STRACE(1);
        // ---- Z = R^-1 Sigma;  y = A^-1 s1, V, M --------------------------------------------------------
        for (int i = tid; i < P * DD; i += NL) {
            const int q = i / DD, r = (i - q * DD) / D, cc = i - q * DD - r * D;
            double v = 0.0;
            for (int k = 0; k < D; ++k) v = fma(s_Ri[q * DD + r * D + k], s_Sig[k * D + cc], v);
            s_Z[i] = v;
        }
        for (int i = tid; i < DD; i += NL) {
            const int a = i / D, k = i - a * D;
            double v = 0.0;
            for (int j = 0; j < D; ++j) v = fma(s_Ai[a * DD + k * D + j], s_ms[a * NM + 1 + j], v);
            s_y[a * D + k] = v;
            s_V[k * D + a] = s_cc[a] * v;
        }
        for (int a = tid; a < D; a += NL) s_M[a] = s_cc[a] * s_ms[a * NM];
        sync();
        }
        GPMPC_STRACE(2);
        // ---- Sigma' = Sigma + S + Sigma V + (Sigma V)^T,  mu' = mu + M,  S -= M M^T ------------------
        for (int i = tid; i < DD; i += NL) {
            const int r = i / D, q = i - r * D;
            double acc = s_Sb[i], vb = 0.0;
            for (int k = 0; k < D; ++k) {
                acc = fma(2.0 * s_Sb[r * D + k], s_V[q * D + k], acc);        // (C_bar V^T)[r][q]
                vb = fma(s_Sig[r * D + k], 2.0 * s_Sb[k * D + q], vb);        // V_bar = Sigma C_bar
            }
            Sacc[i] = acc;
            s_Vb[i] = vb;
        }
        for (int a = tid - W1; a < D; a += NL) { if (a < 0) continue;
            double v = mubar[a];
            for (int b = 0; b < D; ++b) v = fma(-2.0 * s_Sb[a * D + b], s_M[b], v);
            s_Mb[a] = v;
        }
        // pairs: RZ = R^-T Z_bar,  Z_bar = 1/2 W_bar P2
        for (int i = tid - W2; i < P * DD; i += NL) { if (i < 0) continue;
            const int q = i / DD, r = (i - q * DD) / D, cc = i - q * DD - r * D;
            int a, b;
            pair_of(q, a, b);
            const double Wb = ((a == b) ? s_Sb[a * D + a] : 2.0 * s_Sb[a * D + b]) * s_rdet[q];
            const double* mo = s_mom + q * NSP;
            double v = 0.0;
            for (int k = 0; k < D; ++k) v = fma(s_Ri[q * DD + k * D + r], mo[1 + DP + sym_index(k, cc, DP)], v);
            s_RZ[i] = 0.5 * Wb * v;
        }
        sync();
        GPMPC_STRACE(3);
        for (int i = tid; i < DD; i += NL) {
            const int a = i / D, k = i - a * D;
            double v = 0.0;
            for (int j = 0; j < D; ++j) v = fma(s_Ai[a * DD + k * D + j], s_Vb[j * D + a], v);
            s_s1b[a * D + k] = s_cc[a] * v;                                      // s1_bar = c A^-1 v_bar
        }
        for (int a = tid - W1; a < D; a += NL) { if (a < 0) continue;
            double v = s_Mb[a] * s_ms[a * NM];
            for (int k = 0; k < D; ++k) v = fma(s_Vb[k * D + a], s_y[a * D + k], v);
            s_cb[a] = v;                                                         // c_bar
            s_s0b[a] = s_Mb[a] * s_cc[a];
        }
        // pairs: K_q = RZ + (coef R^-T - RZ Z^T) diag(dab);  m_q
        if constexpr (DP <= 4) {
            for (int i = tid - W2; i < P * DD; i += NL) { if (i < 0) continue;   // one thread per element, own buffer
                const int q = i / DD, r = (i - q * DD) / D, cc = i - q * DD - r * D;
                int a, b;
                pair_of(q, a, b);
                const double sb = (a == b) ? s_Sb[a * D + a] : 2.0 * s_Sb[a * D + b];
                const double coef = -0.5 * sb * s_mom[q * NSP] * s_rdet[q];
                double rzzt = 0.0;
                for (int l = 0; l < D; ++l) rzzt = fma(s_RZ[q * DD + r * D + l], s_Z[q * DD + cc * D + l], rzzt);
                s_Kq[i] = s_RZ[i] + (coef * s_Ri[q * DD + cc * D + r] - rzzt) * (c_ils2[a * E + cc] + c_ils2[b * E + cc]);
            }
        } else {
            for (int i = tid; i < P * D; i += NL) {                  // one thread per row, in place over RZ (LDS budget)
                const int q = i / D, r = i - q * D;
                int a, b;
                pair_of(q, a, b);
                const double sb = (a == b) ? s_Sb[a * D + a] : 2.0 * s_Sb[a * D + b];
                const double coef = -0.5 * sb * s_mom[q * NSP] * s_rdet[q];
                double row[8];
#pragma unroll
                for (int l = 0; l < 8; ++l) row[l] = (l < D) ? s_RZ[q * DD + r * D + l] : 0.0;
                for (int cc = 0; cc < D; ++cc) {
                    double rzzt = 0.0;
#pragma unroll
                    for (int l = 0; l < 8; ++l)
                        if (l < D) rzzt = fma(row[l], s_Z[q * DD + cc * D + l], rzzt);
                    double self = 0.0;
#pragma unroll
                    for (int l = 0; l < 8; ++l) self = (l == cc) ? row[l] : self;
                    s_RZ[q * DD + r * D + cc] = self + (coef * s_Ri[q * DD + cc * D + r] - rzzt) * (c_ils2[a * E + cc] + c_ils2[b * E + cc]);
                }
            }
        }
        for (int i = tid - W3; i < P * E; i += NL) { if (i < 0) continue;
            const int q = i / E, e = i - q * E;
            int a, b;
            pair_of(q, a, b);
            const double Wb = ((a == b) ? s_Sb[a * D + a] : 2.0 * s_Sb[a * D + b]) * s_rdet[q];
            const double* mo = s_mom + q * NSP;
            double v;
            if (e < D) {
                double zp = 0.0;
                for (int k = 0; k < D; ++k) zp = fma(s_Z[q * DD + e * D + k], mo[1 + k], zp);
                v = Wb * (mo[1 + e] - (c_ils2[a * E + e] + c_ils2[b * E + e]) * zp);
            } else {
                v = Wb * mo[1 + DP + NH + (e - D)];
            }
            s_mq[i] = v;
        }
        sync();
        GPMPC_STRACE(4);
        // ---- mean part: G1, G2, Ge from the stored moments (q_bar_i = -1/2 lb_i (s0_bar + s1_bar . nu_i)) ----
        for (int i = tid; i < D * NG; i += NL) {
            const int a = i / NG, k = i - a * NG;
            const double* ms = s_ms + a * NM;
            const double* Q2 = ms + 1 + D;
            const double* Q3 = Q2 + T2;
            const double* q1e = Q3 + D * T2;
            const double* q2e = q1e + NX;
            double v;
            if (k < D) {
                v = s_s0b[a] * ms[1 + k];
                for (int l = 0; l < D; ++l) v = fma(Q2[sym_index(k, l, D)], s_s1b[a * D + l], v);
            } else if (k < D + DD) {
                const int d1 = (k - D) / D, d2 = (k - D) - d1 * D;
                const int ti = sym_index(d1, d2, D);
                v = s_s0b[a] * Q2[ti];
                for (int r = 0; r < D; ++r) v = fma(Q3[r * T2 + ti], s_s1b[a * D + r], v);
            } else {
                const int x = k - D - DD;
                v = s_s0b[a] * q1e[x];
                for (int r = 0; r < D; ++r) v = fma(q2e[r * NX + x], s_s1b[a * D + r], v);
            }
            s_Gs[i] = -0.5 * v;
        }
        sync();
        GPMPC_STRACE(5);
        for (int i = tid; i < D * DD; i += NL) {
            const int a = i / DD, k = (i - a * DD) / D, l = i - a * DD - k * D;
            const double* G2 = s_Gs + a * NG + D;
            s_Aib[i] = 0.5 * s_cc[a] * (s_Vb[k * D + a] * s_ms[a * NM + 1 + l] + s_Vb[l * D + a] * s_ms[a * NM + 1 + k])
                     + 0.5 * (G2[k * D + l] + G2[l * D + k]);
        }
        for (int i = tid - W1; i < D * E; i += NL) { if (i < 0) continue;
            const int a = i / E, e = i - a * E;
            const double* G1 = s_Gs + a * NG;
            double v;
            if (e < D) {
                double r = 0.0;
                for (int k = 0; k < D; ++k) r = fma(s_Ai[a * DD + e * D + k], G1[k], r);
                v = -(s_ms[a * NM] * s_s1b[a * D + e] + 2.0 * r);
            } else {
                v = -2.0 * c_ils2[a * E + e] * G1[D + DD + (e - D)];
            }
            s_mba[i] = v;
        }
        sync();
        GPMPC_STRACE(6);
        // A_bar = -A^-1 Ai_bar A^-1 - 1/2 c_bar c A^-1
        for (int i = tid; i < D * DD; i += NL) {
            const int a = i / DD, r = (i - a * DD) / D, cc = i - a * DD - r * D;
            const double* Ai = s_Ai + a * DD;
            double v = 0.0;
            for (int k = 0; k < D; ++k) {
                double w = 0.0;
                for (int l = 0; l < D; ++l) w = fma(s_Aib[a * DD + k * D + l], Ai[l * D + cc], w);
                v = fma(Ai[r * D + k], w, v);
            }
            s_Ab[i] = -v - 0.5 * s_cb[a] * s_cc[a] * Ai[r * D + cc];
        }
        sync();
        GPMPC_STRACE(7);
        // ---- assemble (fixed order) -------------------------------------------------------------------
        for (int i = tid; i < DD; i += NL) {
            double v = Sacc[i];
            for (int a = 0; a < D; ++a) v += s_Ab[a * DD + i];
            for (int q = 0; q < P; ++q) v += s_Kq[q * DD + i];
            Sacc[i] = v;
        }
        for (int e = tid - W1; e < E; e += NL) { if (e < 0) continue;
            double v = (e < D) ? mubar[e] : 0.0;
            for (int a = 0; a < D; ++a) v += s_mba[a * E + e];
            for (int q = 0; q < P; ++q) v += s_mq[q * E + e];
            mbar[e] = v;
        }
        sync();
        GPMPC_STRACE(8);
        for (int i = tid; i < DD; i += NL) {
            const int r = i / D, q = i - r * D;
            Sigbar[i] = 0.5 * (Sacc[i] + Sacc[q * D + r]) + 0.5 * (gSig[t * DD + i] + gSig[t * DD + q * D + r]);
        }
        for (int i = tid - W1; i < D; i += NL) if (i >= 0) mubar[i] = mbar[i] + gmu[t * D + i];
        for (int i = tid - W2; i < A; i += NL) if (i >= 0) p.grad[((size_t)c * H + t) * A + i] = mbar[D + i] + gu[t * A + i];
        sync();
        GPMPC_STRACE(9);
    }
    if (p.host_n > 0 && c == 0) {
        // results to the host's pinned mirror (the gradient just stored by other lanes of this workgroup: loads that bypass the L1),
        // made visible to the host before the sequence number it polls
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        sync();
        for (int i = tid; i < p.host_n; i += NL)
            p.host_out[i] = __hip_atomic_load(p.host_src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        sync();
        if (tid == 0) __hip_atomic_store(p.host_flag, p.host_flag_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
#if defined(GPMPC_PROF_ON)
    if (threadIdx.x == 0 && blockIdx.x == 0)
        printf("PROF sweep prologue cycles: constants + trajectory %lld | cost adjoints %lld | state-independent algebra %lld\n", sprof[10], sprof[11], sprof[12]);
    if (threadIdx.x == 0 && blockIdx.x == 0)
        printf("PROF sweep cycles: load %lld | solves %lld | Z,y,V %lld | Sacc,Vb,RZ %lld | s1b,K,mq %lld | G %lld | Aib,mba %lld | Ab %lld | assemble %lld | finish %lld\n",
               sprof[0], sprof[1], sprof[2], sprof[3], sprof[4], sprof[5], sprof[6], sprof[7], sprof[8], sprof[9]);
#endif
}

}  // namespace gpmpc_hip
