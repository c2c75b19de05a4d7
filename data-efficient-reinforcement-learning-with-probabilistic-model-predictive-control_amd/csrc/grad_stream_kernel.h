// grad_stream_kernel.h -- the pairwise moment pass of the analytic gradient for memories whose per-point arrays do not
// fit the LDS (config 4: N = 1000, D = 4 already needs 222 KiB in pair_moments_kernel).
//
// Same quantities, same row records and the same per-element code as pair_moments_kernel (grad_kernels.h), arranged like
// the streaming forward kernel (rollout_stream_kernel.h): nothing per-point is materialised for all N at once.
//   * mean part: the points go through LDS in chunks of 512 (nu, x_extra, lb_a); the third-order moment sums of a chunk
//     are added to per-(a, component) partial sums that stay in LDS;
//   * pairs, one at a time: the N column factors in LDS; row records produced on the fly for 64-row chunks into a
//     double-buffered stage; a wavefront owns one or two 64-column blocks per pass and keeps their moment accumulators
//     (1 + D + D (D + 1) / 2 + NX per column) in registers across ALL row chunks, so the per-item partial-sum array of the
//     LDS-resident kernel disappears; more column blocks than 8 wavefronts x blocks-per-wave are handled in further passes
//     over the rows.
// LDS use is 8 N bytes + a few tens of KiB: N is bounded by the column-factor array only (~15 000 at D = 4).
#pragma once
#include "grad_kernels.h"

namespace gpmpc_hip {

constexpr int kGsThreads = 512;
constexpr int kGsPB = 512;           // points per chunk of the mean part

struct GsLayout {
    int c_ils2, c_xr, c_logvar, c_tab, m, Sig, aug, ints, nu, xe, lb, mm, kb, stage, part, total;
};

__host__ __device__ inline GsLayout make_gs_layout(int N, int D, int E, int DP, int NXP, int RS, int NSP) {
    GsLayout L;
    const int NX = E - D;
    int o = 0;
    L.c_ils2 = o;   o += rnd2(D * E);
    L.c_xr = o;     o += rnd2(2 * E);
    L.c_logvar = o; o += rnd2(D);
    L.c_tab = o;    o += 64;
    L.m = o;        o += rnd2(E);
    L.Sig = o;      o += rnd2(D * D);
    L.aug = o;      o += (D + 1) * 2 * D * D;
    L.ints = o;     o += 4;
    L.nu = o;       o += DP * kGsPB;
    L.xe = o;       o += NXP * kGsPB;
    L.lb = o;       o += kGsPB;
    L.mm = o;       o += rnd2(D * mean_moment_count(D, NX));
    L.kb = o;       o += rnd2(N);
    L.stage = o;    o += 2 * 64 * RS;
    L.part = o;     o += rnd2((kGsThreads / 64) * NSP);
    L.total = o;
    return L;
}

template <int DP, int NXP>
__global__ __launch_bounds__(kGsThreads) void pair_moments_stream_kernel(const GradArgs p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int NT = kGsThreads;
    constexpr int NW = NT / kWave;
    constexpr int RS = grad_row_stride(DP, NXP);     // row record: ka'_i, beta_ai, g_i (DP), u_i (DP), nu_ie / l_ae^2 (NXP), pad
    constexpr int NH = DP * (DP + 1) / 2;
    constexpr int NSP = 1 + DP + NH + NXP;
    constexpr int CBW = (DP <= 4) ? 2 : 1;   // column blocks per wavefront and pass (accumulators in registers)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = blockIdx.x, c = blockIdx.y;
    const int N = p.N, D = p.D, A = p.A, E = p.E, H = p.H;
    const int NX = E - D;
    const int P = D * (D + 1) / 2;
    const int LD = 2 * D;
    const GsLayout L = make_gs_layout(N, D, E, DP, NXP, RS, NSP);
    double* c_ils2 = smem + L.c_ils2;
    double* c_xr = smem + L.c_xr;
    double* c_logvar = smem + L.c_logvar;
    double* c_tab = smem + L.c_tab;
    double* s_m = smem + L.m;
    double* s_Sig = smem + L.Sig;
    double* s_aug = smem + L.aug;
    int* s_K = reinterpret_cast<int*>(smem + L.ints);
    double* a_nu = smem + L.nu;
    double* a_xe = smem + L.xe;
    double* a_lb = smem + L.lb;
    double* s_mm = smem + L.mm;
    double* a_kb = smem + L.kb;
    double* s_stage = smem + L.stage;
    double* s_part = smem + L.part;

    for (int i = tid; i < D; i += NT) c_logvar[i] = p.logvar[i];
    for (int i = tid; i < D * E; i += NT) c_ils2[i] = p.ils2[i];
    for (int i = tid; i < 64; i += NT) c_tab[i] = kExp2Tab[i];
    for (int i = tid; i < 2 * E; i += NT) c_xr[i] = p.xrange[i];
    for (int i = tid; i < E; i += NT) {
        double v;
        if (i < D) v = p.mu[((size_t)c * (H + 1) + t) * D + i];
        else if (i < D + A) v = p.actions[((size_t)c * H + t) * A + (i - D)];
        else v = p.time0 + (double)t;
        s_m[i] = v;
    }
    for (int i = tid; i < D * D; i += NT) s_Sig[i] = p.Sig[((size_t)c * (H + 1) + t) * D * D + i];
    const int NM = mean_moment_count(D, NX);
    for (int i = tid; i < D * NM; i += NT) s_mm[i] = 0.0;
    for (int i = tid; i < NW * NSP; i += NT) s_part[i] = 0.0;
    __syncthreads();

    // ---- mean part (gp_model.py:140-153): A_a^-1, then lb_ai and the moments of nu under lb_a, chunk by chunk ------------
    // (p.mean_done: written by mean_moments_kernel below)
    if (blockIdx.z == 0 && !p.mean_done) {
        if (tid < D) {
            const int a = tid;
            double* aug = s_aug + a * (D * LD);
            for (int i = 0; i < D; ++i)
                for (int j = 0; j < D; ++j) {
                    aug[i * LD + j] = s_Sig[i * D + j] + (i == j ? 1.0 / c_ils2[a * E + i] : 0.0);
                    aug[i * LD + D + j] = (i == j ? 1.0 : 0.0);
                }
            (void)gauss_solve(aug, D, D, LD);
        }
        __syncthreads();
        for (int c0 = 0; c0 < N; c0 += kGsPB) {
            const int n = (N - c0 < kGsPB) ? (N - c0) : kGsPB;
            for (int i = tid; i < D * n; i += NT) { const int d = i / n, k = i - d * n; a_nu[d * kGsPB + k] = p.Xt[(size_t)d * N + c0 + k] - s_m[d]; }
            for (int i = tid; i < NX * n; i += NT) { const int x = i / n, k = i - x * n; a_xe[x * kGsPB + k] = p.Xt[(size_t)(D + x) * N + c0 + k] - s_m[D + x]; }
            __syncthreads();
            for (int a = 0; a < D; ++a) {
                const double* Ai = s_aug + a * (D * LD) + D;
                for (int k = tid; k < n; k += NT) {
                    double q = 0.0;
                    for (int i = 0; i < D; ++i) {
                        double r = 0.0;
                        for (int j = 0; j < D; ++j) r = fma(Ai[i * LD + j], a_nu[j * kGsPB + k], r);
                        q = fma(a_nu[i * kGsPB + k], r, q);
                    }
                    for (int x = 0; x < NX; ++x) { const double v = a_xe[x * kGsPB + k]; q = fma(v * v, c_ils2[a * E + D + x], q); }
                    a_lb[k] = exp(-0.5 * q) * p.beta[(size_t)a * N + c0 + k];
                }
                __syncthreads();
                for (int comp = wave; comp < NM; comp += NW) {
                    int i1, i2, i3;
                    decode_mean_moment(comp, D, NX, i1, i2, i3);
                    const double* f1 = i1 < 0 ? nullptr : (i1 < D ? a_nu + i1 * kGsPB : a_xe + (i1 - D) * kGsPB);
                    const double* f2 = i2 < 0 ? nullptr : (i2 < D ? a_nu + i2 * kGsPB : a_xe + (i2 - D) * kGsPB);
                    const double* f3 = i3 < 0 ? nullptr : (i3 < D ? a_nu + i3 * kGsPB : a_xe + (i3 - D) * kGsPB);
                    double v = 0.0;
                    if (!f1) { for (int k = lane; k < n; k += 64) v += a_lb[k]; }
                    else if (!f2) { for (int k = lane; k < n; k += 64) v = fma(a_lb[k], f1[k], v); }
                    else if (!f3) { for (int k = lane; k < n; k += 64) v = fma(a_lb[k] * f1[k], f2[k], v); }
                    else { for (int k = lane; k < n; k += 64) v = fma(a_lb[k] * f1[k] * f2[k], f3[k], v); }
                    v = wave_sum(v);
                    if (lane == 0) s_mm[a * NM + comp] += v;             // only this wavefront touches (a, comp)
                }
                __syncthreads();
            }
        }
        for (int i = tid; i < D * NM; i += NT) p.msum[(((size_t)c * H + t) * D) * NM + i] = s_mm[i];
    }

    // ---- pairs (gp_model.py:156-178), one at a time -------------------------------------------------------------------------------
    const int NCB = (N + 63) >> 6;
    const int RC = NCB;                        // 64-row chunks
    for (int q = (int)blockIdx.z; q < P; q += (int)gridDim.z) {
        if (p.sepdone && p.sepdone[((size_t)c * H + t) * P + q]) continue;      // written by sep_grad_moments_kernel
        int a = 0, qq = q;
        while (qq >= D - a) { qq -= D - a; ++a; }
        const int b = a + qq;
        const bool diag = (a == b);
        double* aug = s_aug + D * (D * LD);
        if (tid == 0) {
            for (int i = 0; i < D; ++i)
                for (int j = 0; j < D; ++j) {
                    const double dab = c_ils2[a * E + j] + c_ils2[b * E + j];
                    aug[i * LD + j] = s_Sig[i * D + j] * dab + (i == j ? 1.0 : 0.0);
                    aug[i * LD + D + j] = s_Sig[i * D + j];
                }
            (void)gauss_solve(aug, D, D, LD);
            double cmax = 0.0;
            for (int i = 0; i < D; ++i) {
                const double ui = fmax(fabs(c_xr[i] - s_m[i]), fabs(c_xr[E + i] - s_m[i])) * c_ils2[a * E + i];
                for (int j = 0; j < D; ++j) {
                    const double wj = fmax(fabs(c_xr[j] - s_m[j]), fabs(c_xr[E + j] - s_m[j])) * c_ils2[b * E + j];
                    cmax = fma(fabs(aug[i * LD + D + j]) * ui, wj, cmax);
                }
            }
            int K = 0;
            if (p.force_path != 1 && cmax <= kTaylorMaxArg[kMaxTaylor]) {
                K = 1;
                for (int k = 1; k < kMaxTaylor; ++k) K += (cmax > kTaylorMaxArg[k]) ? 1 : 0;
            }
            s_K[0] = K;
        }
        __syncthreads();
        const int K = __builtin_amdgcn_readfirstlane(s_K[0]);
        const double* Z = aug + D;

        // column factors
        for (int j = tid; j < N; j += NT) {
            double w[DP];
            double ksb = 0.0;
#pragma unroll
            for (int d = 0; d < DP; ++d) {
                const double nu = (d < D) ? p.Xt[(size_t)d * N + j] - s_m[d] : 0.0;
                w[d] = (d < D) ? nu * c_ils2[b * E + d] : 0.0;
                ksb = fma(nu, w[d], ksb);
            }
            for (int x = 0; x < NX; ++x) { const double v = p.Xt[(size_t)(D + x) * N + j] - s_m[D + x]; ksb = fma(v * v, c_ils2[b * E + D + x], ksb); }
            double qb = 0.0;
#pragma unroll
            for (int i = 0; i < DP; ++i)
                if (i < D) {
                    double zw = 0.0;
#pragma unroll
                    for (int jj = 0; jj < DP; ++jj)
                        if (jj < D) zw = fma(Z[i * LD + jj], w[jj], zw);
                    qb = fma(w[i], zw, qb);
                }
            const double kb = c_logvar[b] - 0.5 * ksb + 0.5 * qb;
            // (for a diagonal pair the row factor exp(ka') equals the column factor exp(kb'))
            a_kb[j] = (K > 0) ? (diag ? 2.0 * exp(kb) : exp(kb) * p.beta[(size_t)b * N + j]) : kb;
        }
        __syncthreads();

        // row records of one 64-row chunk; 8 threads per row: thread `part` forms g_part and stores u_part
        auto fill_stage = [&](int r, double* stage) {
            const int trow = tid >> 3, part = tid & 7;
            const int i = r * 64 + trow;
            double gpart = 0.0, upart = 0.0, ks = 0.0, qa = 0.0;
            double* rec = stage + (size_t)trow * RS;
            if (i < N) {
#pragma unroll
                for (int d = 0; d < DP; ++d)
                    if (d < D) {
                        const double nu = p.Xt[(size_t)d * N + i] - s_m[d];
                        const double u = nu * c_ils2[a * E + d];
                        ks = fma(nu, u, ks);
                        if (part < D) gpart = fma(Z[d * LD + part], u, gpart);            // g = Z^T u
                        if (d == part) upart = u;
                    }
                for (int x = 0; x < NX; ++x) {
                    const double v = p.Xt[(size_t)(D + x) * N + i] - s_m[D + x];
                    const double ia = c_ils2[a * E + D + x];
                    ks = fma(v * v, ia, ks);
                    if (x == part) rec[2 + 2 * DP + x] = v * ia;
                }
                qa = upart * gpart;
            }
            qa += __shfl_xor(qa, 1, 64);
            qa += __shfl_xor(qa, 2, 64);
            qa += __shfl_xor(qa, 4, 64);
            if (i >= N) {
                for (int k = part; k < RS; k += 8) rec[k] = 0.0;                           // zero records past the data
            } else {
                if (part < DP) { rec[2 + part] = (part < D) ? gpart : 0.0; rec[2 + DP + part] = (part < D) ? upart : 0.0; }
                for (int x = NX + part; x < NXP; x += 8) rec[2 + 2 * DP + x] = 0.0;
                if (part == 0) {
                    const double ka = c_logvar[a] - 0.5 * ks + 0.5 * qa;
                    const double ba = p.beta[(size_t)a * N + i];
                    if (K > 0) { const double ea = exp(ka); rec[0] = ea; rec[1] = ea * ba; } else { rec[0] = ka; rec[1] = ba; }
                }
            }
        };

        const int npass = (NCB + NW * CBW - 1) / (NW * CBW);
        for (int pass = 0; pass < npass; ++pass) {
            int cb[CBW];
            double cs[CBW], h[CBW][DP + NXP], hh[CBW][NH], w[CBW][DP], kbj[CBW];
            bool vcol[CBW];
#pragma unroll
            for (int s = 0; s < CBW; ++s) {
                cb[s] = (pass * CBW + s) * NW + wave;
                const int j = cb[s] * 64 + lane;
                vcol[s] = (cb[s] < NCB) && (j < N);
                const int jc = vcol[s] ? j : N - 1;
                cs[s] = 0.0;
#pragma unroll
                for (int d = 0; d < DP + NXP; ++d) h[s][d] = 0.0;
#pragma unroll
                for (int k = 0; k < NH; ++k) hh[s][k] = 0.0;
#pragma unroll
                for (int d = 0; d < DP; ++d) w[s][d] = (d < D && vcol[s]) ? (p.Xt[(size_t)d * N + jc] - s_m[d]) * c_ils2[b * E + d] : 0.0;
                kbj[s] = vcol[s] ? a_kb[jc] : 0.0;
            }
            __syncthreads();                                                 // column factors complete; previous pass done with the stage
            fill_stage(0, s_stage);
            __syncthreads();
            for (int r = 0; r < RC; ++r) {
                if (r + 1 < RC) fill_stage(r + 1, s_stage + ((r + 1) & 1) * 64 * RS);
                const double* rec0 = s_stage + (r & 1) * 64 * RS;
                int nch = N - r * 64;
                if (nch > 64) nch = 64;
                nch = (nch + 1) & ~1;                                        // rows past the data are zero records
#pragma unroll
                for (int s = 0; s < CBW; ++s) {
                    if (cb[s] >= NCB) continue;                              // wave-uniform
                    const int j0 = cb[s] * 64;
                    if (diag && j0 + 63 < r * 64) continue;                  // T is zero below its diagonal
                    int nrows = nch;
                    if (diag) { const int lim = (j0 + 64 - r * 64 + 1) & ~1; if (lim < nrows) nrows = lim; }
                    const int jc = vcol[s] ? j0 + lane : N - 1;
                    const double* rec = rec0;
                    const double* Tp = p.Tm + ((size_t)a * (N + kTPad) + (size_t)r * 64) * N + jc;
                    auto accumulate = [&](double e, const double* rr) {
                        cs[s] += e;
                        int k = 0;
#pragma unroll
                        for (int d = 0; d < DP; ++d) {
                            const double td = e * rr[2 + DP + d];
                            h[s][d] += td;
#pragma unroll
                            for (int d2 = d; d2 < DP; ++d2) { hh[s][k] = fma(td, rr[2 + DP + d2], hh[s][k]); ++k; }
                        }
#pragma unroll
                        for (int x = 0; x < NXP; ++x) h[s][DP + x] = fma(e, rr[2 + 2 * DP + x], h[s][DP + x]);
                    };
                    auto run = [&](auto kc) {
                        constexpr int KK = decltype(kc)::value;
                        double tn[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) tn[u] = diag ? Tp[(size_t)u * N] : 1.0;
                        for (int it = 0; it < nrows; it += 2) {
                            double e[2];
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                const double* rr = rec + u * RS;
                                const double tv = tn[u];
                                tn[u] = diag ? Tp[(size_t)(2 + u) * N] : 1.0;
                                if constexpr (KK > 0) {
                                    double cc = rr[2] * w[s][0];
#pragma unroll
                                    for (int d = 1; d < DP; ++d) cc = fma(rr[2 + d], w[s][d], cc);
                                    e[u] = taylor_exp<KK>(cc) * (diag ? rr[0] * tv : rr[1]);
                                } else {
                                    double arg = rr[0] + kbj[s];
#pragma unroll
                                    for (int d = 0; d < DP; ++d) arg = fma(rr[2 + d], w[s][d], arg);
                                    e[u] = fast_exp(arg, c_tab) * (diag ? tv : rr[1]);
                                }
                            }
#pragma unroll
                            for (int u = 0; u < 2; ++u) accumulate(e[u], rec + u * RS);
                            rec += 2 * RS;
                            Tp += (size_t)2 * N;
                        }
                    };
                    if (K == 0) run(std::integral_constant<int, 0>{});
                    else if (K <= 2) run(std::integral_constant<int, 2>{});
                    else if (K <= 4) run(std::integral_constant<int, 4>{});
                    else if (K <= 6) run(std::integral_constant<int, 6>{});
                    else if (K <= 8) run(std::integral_constant<int, 8>{});
                    else if (K <= 10) run(std::integral_constant<int, 10>{});
                    else if (K <= 12) run(std::integral_constant<int, 12>{});
                    else run(std::integral_constant<int, 14>{});
                }
                __syncthreads();
            }
            // fold with the column's own factors and add to this wavefront's partial sums
#pragma unroll
            for (int s = 0; s < CBW; ++s) {
                if (cb[s] >= NCB) continue;
                const int jc = vcol[s] ? cb[s] * 64 + lane : N - 1;
                const double colf = vcol[s] ? (K > 0 ? kbj[s] : (diag ? 2.0 : p.beta[(size_t)b * N + jc])) : 0.0;
                double xb[NXP];
#pragma unroll
                for (int x = 0; x < NXP; ++x) xb[x] = (x < NX && vcol[s]) ? (p.Xt[(size_t)(D + x) * N + jc] - s_m[D + x]) * c_ils2[b * E + D + x] : 0.0;
                double* out = s_part + wave * NSP;
                auto emit = [&](int k, double v) {
                    v = wave_sum(v * colf);
                    if (lane == 0) out[k] += v;
                };
                emit(0, cs[s]);
                int k = 0;
#pragma unroll
                for (int d = 0; d < DP; ++d) {
                    emit(1 + d, h[s][d] + cs[s] * w[s][d]);
#pragma unroll
                    for (int d2 = d; d2 < DP; ++d2) {
                        emit(1 + DP + k, hh[s][k] + cs[s] * w[s][d] * w[s][d2] + h[s][d] * w[s][d2] + w[s][d] * h[s][d2]);
                        ++k;
                    }
                }
#pragma unroll
                for (int x = 0; x < NXP; ++x) emit(1 + DP + NH + x, h[s][DP + x] + cs[s] * xb[x]);
            }
        }
        __syncthreads();
        for (int k = tid; k < NSP; k += NT) {
            double v = 0.0;
            for (int wv = 0; wv < NW; ++wv) { v += s_part[wv * NSP + k]; s_part[wv * NSP + k] = 0.0; }
            p.mom[((((size_t)c * H + t) * P) + q) * NSP + k] = v;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// The mean part of the moment pass on its own (D <= 4): per (candidate, step) and output a the sums over the memory points of
// lb_ai x {1, nu_d, nu_d nu_e, nu_k nu_d nu_e, nu_x, nu_k nu_x} (mean_moment_count: 65 numbers at D = 4 with two action inputs).
// pair_moments_stream_kernel forms them component by component -- one wavefront per component, four LDS reads per point and
// term, a wave reduction per component and 512-point chunk: 22 ms of a config-4 gradient launch whose pairs are all handled by
// the matrix-core and tile kernels (profiles/r04c_c4_gradient_kernel_trace_stats.txt), 20 x what the arithmetic needs.  Here a
// wavefront owns one output a, LANES OWN POINTS: nu, lb and every product stay in registers, the components accumulate in
// registers over the lane's N / 64 points, and ONE round of wave reductions (16 values per 57 instructions: wave_reduce16) ends
// the item.  Grid (H, B), D wavefronts per workgroup.
template <int DP, int NXP>
__device__ __forceinline__ void mean_moments_body(const GradArgs& p, double* s_tab) {
    constexpr int T2 = DP * (DP + 1) / 2;
    constexpr int NC = 1 + DP + T2 + DP * T2 + NXP + DP * NXP;       // components in the compile-time (DP, NXP) layout
    constexpr int NG = (NC + 15) / 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int a = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = blockIdx.x, c = blockIdx.y;
    const int N = p.N, D = p.D, A = p.A, E = p.E, H = p.H, NX = E - D;
    if (tid < 64) s_tab[tid] = kExp2Tab[tid];
    __syncthreads();
    if (a >= D) return;
    // input mean of the step and A_a^-1 = (Sigma + diag(l_a^2))^-1 (every lane redundantly: wave-uniform operands)
    double md[DP], mx[NXP], ilx[NXP];
#pragma unroll
    for (int d = 0; d < DP; ++d) md[d] = (d < D) ? p.mu[((size_t)c * (H + 1) + t) * D + d] : 0.0;
#pragma unroll
    for (int x = 0; x < NXP; ++x) {
        const int e = D + x;
        mx[x] = (x >= NX) ? 0.0 : (e < D + A ? p.actions[((size_t)c * H + t) * A + (e - D)] : p.time0 + (double)t);
        ilx[x] = (x < NX) ? p.ils2[(size_t)a * E + e] : 0.0;
    }
    double m[DP][2 * DP];
    {
        const double* Sg = p.Sig + ((size_t)c * (H + 1) + t) * D * D;
#pragma unroll
        for (int i = 0; i < DP; ++i)
#pragma unroll
            for (int j = 0; j < DP; ++j) {
                const bool in = (i < D && j < D);
                m[i][j] = (in ? Sg[i * D + j] : 0.0) + (i == j ? (i < D ? 1.0 / p.ils2[(size_t)a * E + i] : 1.0) : 0.0);
                m[i][DP + j] = (i == j) ? 1.0 : 0.0;
            }
        (void)small_solve<DP>(m);
    }
    double acc[NG * 16];
#pragma unroll
    for (int k = 0; k < NG * 16; ++k) acc[k] = 0.0;
    const double* beta = p.beta + (size_t)a * N;
    for (int pt = lane; pt < N; pt += 64) {
        double nu[DP], xe[NXP];
#pragma unroll
        for (int d = 0; d < DP; ++d) nu[d] = (d < D) ? p.Xt[(size_t)d * N + pt] - md[d] : 0.0;
#pragma unroll
        for (int x = 0; x < NXP; ++x) xe[x] = (x < NX) ? p.Xt[(size_t)(D + x) * N + pt] - mx[x] : 0.0;
        double q = 0.0;
#pragma unroll
        for (int i = 0; i < DP; ++i) {
            double r = 0.0;
#pragma unroll
            for (int j = 0; j < DP; ++j) r = fma((i < D && j < D) ? m[i][DP + j] : 0.0, nu[j], r);
            q = fma(nu[i], r, q);
        }
#pragma unroll
        for (int x = 0; x < NXP; ++x) q = fma(xe[x] * xe[x], ilx[x], q);
        const double lb = fast_exp(-0.5 * q, s_tab) * beta[pt];                  // gp_model.py:148
        int n = 0;
        acc[n++] += lb;
        double l1[DP];
#pragma unroll
        for (int d = 0; d < DP; ++d) { l1[d] = lb * nu[d]; acc[n++] += l1[d]; }
        double l2[T2];
        {
            int k = 0;
#pragma unroll
            for (int d1 = 0; d1 < DP; ++d1)
#pragma unroll
                for (int d2 = d1; d2 < DP; ++d2) { l2[k] = l1[d1] * nu[d2]; acc[n++] += l2[k]; ++k; }
        }
#pragma unroll
        for (int k = 0; k < DP; ++k)
#pragma unroll
            for (int tt = 0; tt < T2; ++tt) { acc[n] = fma(l2[tt], nu[k], acc[n]); ++n; }
#pragma unroll
        for (int x = 0; x < NXP; ++x) { acc[n] = fma(lb, xe[x], acc[n]); ++n; }
#pragma unroll
        for (int k = 0; k < DP; ++k)
#pragma unroll
            for (int x = 0; x < NXP; ++x) { acc[n] = fma(l1[k], xe[x], acc[n]); ++n; }
    }
    // one round of reductions: lane 4 m of group g holds the total of component 16 g + m of the (DP, NXP) layout; its slot in the
    // (D, NX) layout of msum comes from the component's factors
    const int NM = mean_moment_count(D, NX);
    double* out = p.msum + (((size_t)c * H + t) * D + a) * NM;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const double(&grp)[16] = *reinterpret_cast<const double(*)[16]>(&acc[16 * g]);
        const double tot = wave_reduce16(grp);
        const int comp = 16 * g + (lane >> 2);
        if ((lane & 3) == 0 && comp < NC) {
            int i1, i2, i3;
            decode_mean_moment(comp, DP, NXP, i1, i2, i3);                 // factors: < DP a state dimension, DP + x an extra input
            auto state_ok = [&](int i) { return i < 0 || (i < DP ? i < D : i - DP < NX); };
            if (state_ok(i1) && state_ok(i2) && state_ok(i3)) {
                const int T2r = tri_count(D);
                int slot;
                if (i1 < 0) slot = 0;
                else if (i2 < 0) slot = (i1 < DP) ? 1 + i1 : 1 + D + T2r + D * T2r + (i1 - DP);
                else if (i3 < 0) slot = (i2 < DP) ? 1 + D + tri_index(i1, i2, D) : 1 + D + T2r + D * T2r + NX + i1 * NX + (i2 - DP);
                else slot = 1 + D + T2r + i1 * T2r + tri_index(i2, i3, D);
                out[slot] = tot;
            }
        }
    }
}

template <int DP, int NXP>
__global__ __launch_bounds__(64 * DP) void mean_moments_kernel(const GradArgs p) {
    __shared__ double s_tab[64];
    mean_moments_body<DP, NXP>(p, s_tab);
}


}  // namespace gpmpc_hip
