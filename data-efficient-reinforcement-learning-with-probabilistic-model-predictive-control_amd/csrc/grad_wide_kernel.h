// grad_wide_kernel.h -- analytic gradient dJ/du for state dimensions beyond 8 (config 5: D = 16), gfx950 / MI355X.
//
// The reference obtains the gradient by torch autograd through predict_trajectory (gp_mpc_controller.py:277,
// `mean_cost.backward()`).  grad_kernels.h covers D <= 8 with per-column register accumulators of every moment
// (1 + 2 D + D (D + 1) / 2 + A per column): at D = 16 that is 169 accumulators per column, and the reverse sweep's
// per-pair LDS arrays would be 278 KiB.  This file is the same algebra (the test suite states it in numpy: the adjoint checker) arranged
// for wide states:
//
//  wide_pair_moments_kernel   one workgroup per (output pair, horizon step, candidate).  The pairwise weights
//      E_ij = (beta_ai beta_bj - [a = b] iK_a,ij) exp(ka'_i + kb'_j + u_i^T Z w_j)   (gp_model.py:161-175)
//      are formed 16 x 16 tiles at a time on the fp64 matrix cores: c = u (Z w)^T is 4 v_mfma_f64_16x16x4_f64 with the
//      Z-transformed vector on the COLUMN side (h_j = Z w_j, once per 16 columns), so the row operand is the plain
//      u_i = nu_i / l_a^2 read straight from X -- nothing per-point is staged.  Only two things are accumulated per
//      element: the column sum c_j = sum_i E_ij (a VALU add) and V_j = sum_i E_ij u_i, which is 4 more MFMAs whose A operand
//      is the E tile AS IT LIES in the accumulator layout of the first product (lane l, register r = row 4 r + (l >> 4),
//      column l & 15 = the transposed A-operand layout), B = u rows.  Per 16 columns the moments then follow from
//      O(16 D^2) work:  W += c_j,  P1 += V_j + c_j w_j,  P2 += c_j w_j w_j^T + V_j w_j^T + w_j V_j^T,  Pe += c_j nu_jx / l_bx^2.
//      The row-side terms (sum_i r_i u_i u_i^T, r_i = sum_j E_ij) are the column-side terms of the pair evaluated in the
//      other orientation (rows and columns swapped, Z transposed): a second, lighter pass without the V product;
//      a diagonal pair is symmetric and takes one pass over the full square with weights from the symmetric iK.
//  wide_adjoint_sweep_kernel  one workgroup per candidate, t = H-1 .. 0: the D x D algebra of the reverse step
//      backward_step with one pair per wavefront at a time (its 16 x 16 blocks in that wavefront's LDS scratch), and the mean
//      part's point passes (lb_i, then the adjoint weights om_i and their moments G1, G2, Ge) in chunks of 256 points
//      staged through LDS, every thread owning one entry of G2.
// Direct exp everywhere (table-based fast_exp): this path is about having the analytic gradient at D = 16 at ~4 forward
// rollouts per gradient instead of 4 H A + 1 = 801 differenced ones; fixed summation order, bitwise reproducible.
#pragma once
#include "rollout_stream_kernel.h"
#include "grad_kernels.h"

namespace gpmpc_hip {

constexpr int kWideThreads = 512;      // moment pass: 8 wavefronts per (pair, step, candidate), two per SIMD
constexpr int kWideSweepThreads = 512;     // 256 VGPRs per thread: at 1024 threads the kernel spilled 46 VGPRs and 204 SGPRs
constexpr int kWidePairWaves = 4;      // sweep: wavefronts that work on pairs concurrently (LDS scratch each)

struct WideArgs {
    const double* Xt;       // (E, N)
    const double* beta;     // (D, N)
    const double* iK;       // (D, N, N) symmetric
    const double* ils2;     // (D, E)
    const double* var;      // (D)
    const double* logvar;   // (D)
    const double* cost;     // target | W | W_T | smin | smax
    const double* actions;  // (B, H, A)
    const double* mu;       // (B, H + 1, D)
    const double* Sig;      // (B, H + 1, D, D)
    const double* cv;       // (B, H + 1)
    double* mom;            // (B, H, P, NSP): W | P1 (D) | P2 (D x D) | Pe (NX)
    double* grad;           // (B, H, A)
    double kappa;
    int use_constraints;
    int N, D, A, E, H, B;
    int include_time;
    double time0;
    int NSP;
    const double* xrange;   // (2, E) per-dimension min | max of the memory points
    int force_path;         // 1: direct exp everywhere (tests)
};

__host__ __device__ inline int wide_nsp(int D, int NX) { return rnd2(1 + D + D * D + NX); }

constexpr int kWideStageRows = 64;     // rows per staged chunk (two buffers)
constexpr int kWideRS = 18;            // row record: factor | beta | u (16)  (an odd stride removes the A-operand bank conflicts and is 3 % slower:
                                       // rollout_stream_kernel.h stream_row_stride, profiles/r04c_ab_c5_grad_*.txt)
// per wavefront: V tile | w tile | c_j | extra inputs of the 16 columns (stride 8 up to 8 extra inputs, else 16) | column factors
__host__ __device__ inline int wide_fx_stride(int NX) { return NX <= 8 ? 8 : 16; }
__host__ __device__ inline int wide_fold_words(int NX) { return 2 * 256 + 16 + 16 * wide_fx_stride(NX) + 16; }
constexpr int kWideRowSlots = 2 * (kWideThreads / 64) * kWideStageRows;     // row sums of a chunk per wavefront, double-buffered

struct WideMomLayout {
    int aug, Z, m, ila, ilb, ka, kb, exptab, fold, rs, tot, etab, stage, flag, total;
};

__host__ __device__ inline WideMomLayout make_wide_mom_layout(int N, int D, int E, int NSP) {
    WideMomLayout L;
    int o = 0;
    L.exptab = o; o += 64;
    L.aug = o;    o += 2 * D * D;
    L.Z = o;      o += 2 * 16 * 16;              // Z and Z^T, padded to 16 x 16
    L.m = o;      o += rnd2(E);
    L.ila = o;    o += rnd2(E);
    L.ilb = o;    o += rnd2(E);
    L.ka = o;     o += rnd2(N);
    L.kb = o;     o += rnd2(N);
    L.fold = o;   o += (kWideThreads / 64) * wide_fold_words(E - D);
    L.rs = o;     o += (E - D <= 8) ? kWideRowSlots : 0;       // (the space the narrower extra-input tile frees)
    L.etab = o;   o += 2 * kTableHalf + 2;                    // exp(n / 128), |n| <= 1024: the tabulated form of the forward kernel
    // row records of a 64-row chunk, double-buffered; the per-wavefront totals of the epilogue reuse the region
    const int st = 2 * kWideStageRows * kWideRS, tt = (kWideThreads / 64) * NSP;
    L.stage = o;  L.tot = o;  o += rnd2(st > tt ? st : tt);
    L.flag = o;   o += 2;
    L.total = o;
    return L;
}

// ------------------------------------------------------------------------------------------
template <int DP>
__global__ __launch_bounds__(kWideThreads) void wide_pair_moments_kernel(const WideArgs p) {
    static_assert(DP == 16, "the wide gradient path is built on 16 x 16 x 4 fp64 MFMA tiles");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int NT = kWideThreads, NW = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = blockIdx.x, t = blockIdx.y, c = blockIdx.z;
    const int N = p.N, D = p.D, A = p.A, E = p.E, H = p.H;
    const int NX = E - D, LD = 2 * D, NSP = p.NSP;
    int a = 0, rem = q;
    while (rem >= D - a) { rem -= D - a; ++a; }
    const int b = a + rem;
    const bool diag = (a == b);
    const WideMomLayout L = make_wide_mom_layout(N, D, E, NSP);
    double* s_exptab = smem + L.exptab;
    double* s_aug = smem + L.aug;
    double* s_Z = smem + L.Z;                 // [0, 256): Z (row-major, stride 16), [256, 512): Z^T
    double* s_m = smem + L.m;
    double* s_ila = smem + L.ila;
    double* s_ilb = smem + L.ilb;
    double* s_ka = smem + L.ka;
    double* s_kb = smem + L.kb;
    double* s_fold = smem + L.fold + wave * wide_fold_words(NX);
    double* s_rs = smem + L.rs;
    const int FX = wide_fx_stride(NX);
    double* s_tot = smem + L.tot;
    double* s_etab = smem + L.etab + kTableHalf;           // centre of the table
    double* s_stage = smem + L.stage;
    int* s_flag = reinterpret_cast<int*>(smem + L.flag);

    const double* mu = p.mu + ((size_t)c * (H + 1) + t) * D;
    const double* Sg = p.Sig + ((size_t)c * (H + 1) + t) * D * D;
    for (int i = tid; i < 64; i += NT) s_exptab[i] = kExp2Tab[i];
    for (int i = tid; i <= 2 * kTableHalf; i += NT) s_etab[i - kTableHalf] = exp((double)(i - kTableHalf) * 0.0078125);
    for (int e = tid; e < E; e += NT) {
        double v;
        if (e < D) v = mu[e];
        else if (e < D + A) v = p.actions[((size_t)c * H + t) * A + (e - D)];
        else v = p.time0 + (double)t;
        s_m[e] = v;
        s_ila[e] = p.ils2[(size_t)a * E + e];
        s_ilb[e] = p.ils2[(size_t)b * E + e];
    }
    for (int i = tid; i < 512; i += NT) s_Z[i] = 0.0;
    __syncthreads();
    if (wave == 0) {
        // R = Sigma diag(1/l_a^2 + 1/l_b^2) + I,  Z = R^-1 Sigma   (gp_model.py:156-163)
        for (int idx = lane; idx < D * D; idx += 64) {
            const int i = idx / D, j = idx - i * D;
            s_aug[i * LD + j] = Sg[idx] * (s_ila[j] + s_ilb[j]) + (i == j ? 1.0 : 0.0);
            s_aug[i * LD + D + j] = Sg[idx];
        }
        wave_lds_sync();
        (void)wave_gauss_solve(s_aug, D, D, LD, lane);
        for (int idx = lane; idx < D * D; idx += 64) {
            const int i = idx / D, j = idx - i * D;
            const double z = s_aug[i * LD + D + j];
            s_Z[i * 16 + j] = z;
            s_Z[256 + j * 16 + i] = z;
        }
        // |u_i^T Z w_j| <= sum_dd' |Z_dd'| umax_d wmax_d' over the data range of the memory points: within the table's range the
        // pair takes the staged, tabulated form below (as rollout_stream_kernel.h), beyond it the direct exponential
        double cpart = 0.0;
        for (int idx = lane; idx < D * D; idx += 64) {
            const int i = idx / D, j = idx - i * D;
            const double mi = mu[i], mj = mu[j];
            const double ui = fmax(fabs(p.xrange[i] - mi), fabs(p.xrange[E + i] - mi)) * s_ila[i];
            const double wj = fmax(fabs(p.xrange[j] - mj), fabs(p.xrange[E + j] - mj)) * s_ilb[j];
            cpart = fma(fabs(s_aug[i * LD + D + j]) * ui, wj, cpart);
        }
        const double cmax = wave_sum(cpart);
        if (lane == 0) s_flag[0] = (p.force_path != 1 && cmax <= kTableMaxArg) ? 1 : 0;
    }
    __syncthreads();
    const bool use_table = __builtin_amdgcn_readfirstlane(s_flag[0]) != 0;
    // per-point log-factors:  k' = log var - sum_e nu_e^2 / (2 l_e^2) + x^T Z x / 2,  x = nu / l^2 (state part)
    auto kprime = [&](int pt, int side) -> double {
        const double* il = side ? s_ilb : s_ila;
        double x[DP];
        double ks = 0.0;
#pragma unroll
        for (int d = 0; d < DP; ++d) {
            const double nu = (d < D) ? p.Xt[(size_t)d * N + pt] - s_m[d] : 0.0;
            x[d] = (d < D) ? nu * il[d] : 0.0;
            ks = fma(nu, x[d], ks);
        }
        for (int e = D; e < E; ++e) {
            const double v = p.Xt[(size_t)e * N + pt] - s_m[e];
            ks = fma(v * v, il[e], ks);
        }
        double qq = 0.0;
#pragma unroll 4
        for (int i = 0; i < DP; ++i) {
            double zx = 0.0;
#pragma unroll
            for (int j = 0; j < DP; ++j) zx = fma(s_Z[i * 16 + j], x[j], zx);
            qq = fma(x[i], zx, qq);
        }
        return (side ? p.logvar[b] : p.logvar[a]) - 0.5 * ks + 0.5 * qq;
    };
    // Row-sum mode (tabulated form, off-diagonal pair, <= 8 extra inputs): ONE orientation.  The second orientation re-evaluated all N^2
    // weights only for their row sums r_i = sum_j E_ij; here the tiles of the first orientation give them -- every wavefront reduces
    // its tile's rows over the 16 lanes of a DPP row, the eight wavefronts' partials of a 64-row chunk meet in LDS and are added in a
    // fixed order into r (which lives where the column side's log-factors were: those are formed per sweep for the wavefront's 16
    // columns instead).
    const bool rowsum = use_table && !diag && NX <= 8;
    double* s_r = s_kb;
    for (int pt = tid; pt < N; pt += NT) {
        s_ka[pt] = kprime(pt, 0);
        if (rowsum) s_r[pt] = 0.0;
        else s_kb[pt] = diag ? s_ka[pt] : kprime(pt, 1);
    }
    __syncthreads();

    // totals of this wavefront (lanes own entries): P2 entries lane * 4 .. + 3, P1 entry lane (< 16), Pe entry lane (< NX), W lane 0
    double tP2[4] = {0.0, 0.0, 0.0, 0.0}, tP1 = 0.0, tPe = 0.0, tW = 0.0;
    const int NT16 = (N + 15) >> 4;
    const int col16 = lane & 15, grp = lane >> 4;
    double* f_V = s_fold;                  // [col][dim]
    double* f_w = s_fold + 256;            // [col][dim]   column-side monomial vector x_j = nu_j / l^2
    double* f_c = s_fold + 512;            // [col]
    double* f_x = s_fold + 528;            // [col][extra input] nu_jx
    double* f_cf = s_fold + 528 + 16 * FX; // [col] column factor (tabulated form)

    // the 16 columns of a tile folded into the moments (lanes own entries); f_c, f_V, f_w, f_x hold the tile
    auto fold_tile = [&](int orient, const double* il_r, const double* il_c) {
        if (orient == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = lane * 4 + k, m_ = e >> 4, n_ = e & 15;
                double s = 0.0;
                _Pragma("unroll 2") for (int cc = 0; cc < 16; ++cc) {
                    const double cj = f_c[cc], wm = f_w[cc * 16 + m_], wn = f_w[cc * 16 + n_];
                    s = fma(cj * wm, wn, s);
                    s = fma(f_V[cc * 16 + m_], wn, s);
                    s = fma(wm, f_V[cc * 16 + n_], s);
                    if (diag) s = fma(cj * wm, wn, s);            // symmetric pair: the row-side term equals the column-side one
                }
                tP2[k] += s;
            }
            if (lane < 16) {
                double s = 0.0;
                _Pragma("unroll 2") for (int cc = 0; cc < 16; ++cc) s += f_V[cc * 16 + lane] + f_c[cc] * f_w[cc * 16 + lane];
                tP1 += s;
            }
            if (lane < NX) {
                double s = 0.0;
                _Pragma("unroll 2") for (int cc = 0; cc < 16; ++cc) s = fma(f_c[cc], f_x[cc * FX + lane], s);
                tPe += s * (il_c[D + lane] + (diag ? il_r[D + lane] : 0.0));
            }
            if (lane == 0) {
                double s = 0.0;
                _Pragma("unroll 2") for (int cc = 0; cc < 16; ++cc) s += f_c[cc];
                tW += s;
            }
        } else {
            // other orientation: its column sums are the row sums r_i of the pair; columns here are side-a points (u)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = lane * 4 + k, m_ = e >> 4, n_ = e & 15;
                double s = 0.0;
                _Pragma("unroll 2") for (int cc = 0; cc < 16; ++cc) s = fma(f_c[cc] * f_w[cc * 16 + m_], f_w[cc * 16 + n_], s);
                tP2[k] += s;
            }
            if (lane < NX) {
                double s = 0.0;
                _Pragma("unroll 2") for (int cc = 0; cc < 16; ++cc) s = fma(f_c[cc], f_x[cc * FX + lane], s);
                tPe += s * il_c[D + lane];
            }
        }
    };
    // the column tile's vectors: x_j (16 values per column) and the extra inputs -> fold scratch
    auto load_columns = [&](int ct, const double* il_c) {
        for (int k = lane; k < 256; k += 64) {
            const int cc = k >> 4, d = k & 15;
            int jj = ct * 16 + cc;
            jj = jj < N ? jj : N - 1;
            f_w[k] = (d < D) ? (p.Xt[(size_t)d * N + jj] - s_m[d]) * il_c[d] : 0.0;
        }
        for (int k = lane; k < 16 * NX; k += 64) {
            const int cc = k / NX, x = k - cc * NX;
            int jj = ct * 16 + cc;
            jj = jj < N ? jj : N - 1;
            f_x[cc * FX + x] = p.Xt[(size_t)(D + x) * N + jj] - s_m[D + x];
        }
    };

    if (use_table) {
        // ---- staged, tabulated form.  E_ij = wr_i wc_j e^{c_ij} with wr_i = beta_ri e^{k_ri} in the row record and wc_j applied to
        // the column sums (diagonal pairs: (beta_ri beta_cj - iK_ij) e^{k_ri} e^{k_cj} e^{c_ij}); e^c = T[n] P_5(r) as in the forward
        // kernel.  The four wavefronts walk the rows together: 64-row chunks of row records {wr | beta | u (16)} are staged in LDS
        // (double-buffered, one barrier per chunk) and shared by the four column tiles in flight; operands of both matrix products
        // are LDS reads, nothing is fetched from global memory inside the tile loop except iK of a diagonal pair.
        constexpr int RSW = kWideRS, CHW = kWideStageRows;
        const int nchunk = (N + CHW - 1) / CHW;
        const int nsweep = (NT16 + NW - 1) / NW;
        constexpr double kShift = 52776558133248.0;                                     // 1.5 * 2^45: see block_mfma_table
        for (int orient = 0; orient < ((diag || rowsum) ? 1 : 2); ++orient) {
            const int rs = orient ? b : a, cs_ = orient ? a : b;
            const double* il_r = orient ? s_ilb : s_ila;
            const double* il_c = orient ? s_ila : s_ilb;
            const double* k_r = orient ? s_kb : s_ka;
            const double* k_c = orient ? s_ka : s_kb;
            const double* Zu = s_Z + (orient ? 256 : 0);
            const double* beta_r = p.beta + (size_t)rs * N;
            const double* beta_c = p.beta + (size_t)cs_ * N;
            // Stage fill in two halves: the chunk's inputs are LOADED (unconditionally, from clamped indices: a load behind a per-lane
            // condition compiles to a branch with its own wait) before the tile loop of the current chunk and turned into row records
            // AFTER it -- in one piece at the head of the iteration every wavefront of the workgroup sat out the L2 latency together
            // once per 64-row chunk (the chunks are barrier-paced: nothing else runs on the SIMDs meanwhile).
            constexpr int TPR = NT / CHW, DPT = 16 / TPR;          // threads per row, dimensions per thread
            auto fill_load = [&](int ch, double (&fx)[DPT], double& fb) {
                const int row = tid / TPR, part = tid - row * TPR;
                const int i = ch * CHW + row;
                const int ic = i < N ? i : N - 1;
#pragma unroll
                for (int k = 0; k < DPT; ++k) {
                    const int d = part * DPT + k;
                    fx[k] = p.Xt[(size_t)(d < D ? d : D - 1) * N + ic];
                }
                fb = beta_r[ic];
            };
            auto fill_store = [&](int ch, double* st, const double (&fx)[DPT], double fb) {
                const int row = tid / TPR, part = tid - row * TPR;
                const int i = ch * CHW + row;
                const bool in = i < N;
                const int ic = in ? i : N - 1;
                double* rec = st + (size_t)row * RSW;
#pragma unroll
                for (int k = 0; k < DPT; ++k) {
                    const int d = part * DPT + k;
                    const int dc = d < D ? d : 0;
                    rec[2 + d] = (in && d < D) ? (fx[k] - s_m[dc]) * il_r[dc] : 0.0;
                }
                if (part == 0) {
                    const double er = in ? fast_exp(k_r[ic], s_exptab) : 0.0;
                    rec[0] = diag ? er : er * fb;
                    rec[1] = fb;
                }
            };
            auto fill = [&](int ch, double* st) {
                double fx[DPT], fb;
                fill_load(ch, fx, fb);
                fill_store(ch, st, fx, fb);
            };
            for (int sw = 0; sw < nsweep; ++sw) {
                const int ct = sw * NW + wave;
                const bool act = ct < NT16;                       // wave-uniform; idle wavefronts still fill and meet the barriers
                const int j = ct * 16 + col16;
                const bool jin = act && j < N;
                const int jc = (act && j < N) ? j : N - 1;
                double hB[4] = {0.0, 0.0, 0.0, 0.0};
                if (act) {
                    load_columns(ct, il_c);
                    wave_lds_sync();
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const int d = 4 * qd + grp;
                        double h = 0.0;
#pragma unroll
                        for (int e = 0; e < 16; ++e) h = fma(Zu[d * 16 + e], f_w[col16 * 16 + e], h);
                        hB[qd] = h;
                    }
                }
                const double bcj = jin ? beta_c[jc] : 0.0;
                const double ecj = jin ? fast_exp(rowsum ? kprime(jc, 1) : k_c[jc], s_exptab) : 0.0;
                const double colf = diag ? ecj : ecj * bcj;       // column factor of the sums (diagonal pair: beta_cj sits in the weight)
                double csum0 = 0.0, csum1 = 0.0;
                mfma_d4 vc = {0.0, 0.0, 0.0, 0.0};
                __syncthreads();                                   // previous sweep done with the stage
                fill(0, s_stage);
                __syncthreads();
                // row sums of chunk `ch` (written by the wavefronts before the chunk's barrier) -> r.  Wavefront w takes rows 8 w .. 8 w + 7:
                // lane (row = lane >> 3, source wavefront = lane & 7) reads ONE partial at the head of the next chunk's tile loop and the
                // eight partials of a row are added in lane order (a DPP prefix over the lanes of the group) behind it -- a single
                // no-return ds_add_f64 per row and chunk, so r's additions have one fixed order and nothing waits for them.
                auto take_load = [&](int ch) -> double {
                    return s_rs[((ch & 1) * NW + (lane & 7)) * CHW + wave * 8 + (lane >> 3)];
                };
                auto take_add = [&](int ch, double pv) {
                    pv += dpp_shifted<0x111, 0xf>(pv);      // row_shr:1
                    pv += dpp_shifted<0x112, 0xf>(pv);      // row_shr:2
                    pv += dpp_shifted<0x114, 0xf>(pv);      // row_shr:4   -> lane 8 g + 7 holds the sum of lanes 8 g .. 8 g + 7
                    const int row = ch * CHW + wave * 8 + (lane >> 3);
                    if ((lane & 7) == 7 && row < N)
                        (void)__hip_atomic_fetch_add(&s_r[row], pv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                };
                if (rowsum && !act) {                       // an idle wavefront's slots read as zero (both buffers)
                    s_rs[(0 * NW + wave) * CHW + lane] = 0.0;
                    s_rs[(1 * NW + wave) * CHW + lane] = 0.0;
                }
                for (int ch = 0; ch < nchunk; ++ch) {
                    double nfx[DPT], nfb = 0.0;
                    const bool more = ch + 1 < nchunk;
                    if (more) fill_load(ch + 1, nfx, nfb);
                    double prev_rows = 0.0;
                    if (rowsum && ch > 0) prev_rows = take_load(ch - 1);
                    const double* st = s_stage + (ch & 1) * CHW * RSW;
                    if (act) {
#pragma unroll 1
                        for (int rt = 0; rt < CHW / 16; rt += 2) {
                            // two 16-row tiles per iteration: 8 independent evaluations of the exponential per lane
                            const double* a0p = st + (size_t)(16 * rt + col16) * RSW + 2 + grp;
                            const double* a1p = a0p + 16 * RSW;
                            mfma_d4 c0 = {0.0, 0.0, 0.0, 0.0}, c1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                            for (int qd = 0; qd < 4; ++qd) {
                                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0p[4 * qd], hB[qd], c0, 0, 0, 0);
                                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1p[4 * qd], hB[qd], c1, 0, 0, 0);
                            }
                            const double* w0 = st + (size_t)(16 * rt + grp) * RSW;      // row 16 rt + 4 r + grp: + 4 r RSW
                            double wt[8], cv[8], nv[8], tv[8], qv[8];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const double* q0 = w0 + (size_t)(4 * r) * RSW;
                                const double* q1 = q0 + 16 * RSW;
                                if (diag) {
                                    const int i0 = ch * CHW + 16 * rt + 4 * r + grp, i1 = i0 + 16;
                                    const double k0 = p.iK[((size_t)a * N + (i0 < N ? i0 : N - 1)) * N + jc];
                                    const double k1 = p.iK[((size_t)a * N + (i1 < N ? i1 : N - 1)) * N + jc];
                                    wt[r] = q0[0] * fma(q0[1], bcj, -k0);
                                    wt[4 + r] = q1[0] * fma(q1[1], bcj, -k1);
                                } else {
                                    wt[r] = q0[0];
                                    wt[4 + r] = q1[0];
                                }
                                cv[r] = c0[r];
                                cv[4 + r] = c1[r];
                            }
#pragma unroll
                            for (int e = 0; e < 8; ++e) nv[e] = cv[e] + kShift;
#pragma unroll
                            for (int e = 0; e < 8; ++e) tv[e] = s_etab[(int)__double2loint(nv[e])];
#pragma unroll
                            for (int e = 0; e < 8; ++e) cv[e] -= nv[e] - kShift;
#pragma unroll
                            for (int e = 0; e < 8; ++e) qv[e] = fma(cv[e], 1.0 / 120, 1.0 / 24);
#pragma unroll
                            for (int e = 0; e < 8; ++e) qv[e] = fma(qv[e], cv[e], 1.0 / 6);
#pragma unroll
                            for (int e = 0; e < 8; ++e) qv[e] = fma(qv[e], cv[e], 0.5);
#pragma unroll
                            for (int e = 0; e < 8; ++e) qv[e] = fma(qv[e], cv[e], 1.0);
#pragma unroll
                            for (int e = 0; e < 8; ++e) qv[e] = fma(qv[e], cv[e], 1.0);
#pragma unroll
                            for (int e = 0; e < 8; ++e) wt[e] *= tv[e];
#pragma unroll
                            for (int e = 0; e < 8; ++e) wt[e] *= qv[e];               // E_ij without the column factor
                            csum0 += (wt[0] + wt[1]) + (wt[2] + wt[3]);
                            csum1 += (wt[4] + wt[5]) + (wt[6] + wt[7]);
                            if (rowsum) {
                                // sum over the tile's 16 columns (the lanes of a DPP row) of E_ij WITH the column factor, eight rows at a
                                // time on the halving exchanges: lane (bit 3, bit 2) of the row ends with the totals of values 4 b3 + 2 b2, + 1
                                double r4[4], r2[2];
#pragma unroll
                                for (int k = 0; k < 4; ++k) r4[k] = rot_add8(wt[k] * colf, wt[k + 4] * colf);
#pragma unroll
                                for (int k = 0; k < 2; ++k) r2[k] = rot_add4(r4[k], r4[k + 2]);
                                const double t0 = quad_total(r2[0]), t1 = quad_total(r2[1]);
                                if ((col16 & 3) == 0) {
                                    // value index e = 4 b3 + 2 b2 (+ 1): rows 16 rt + 4 e + grp (e < 4), 16 rt + 16 + 4 (e - 4) + grp (e >= 4)
                                    const int e0 = ((col16 >> 3) & 1) * 4 + ((col16 >> 2) & 1) * 2;
                                    double* dst = s_rs + ((ch & 1) * NW + wave) * CHW + 16 * rt + grp;
                                    dst[(e0 < 4) ? 4 * e0 : 16 + 4 * (e0 - 4)] = t0;
                                    dst[(e0 + 1 < 4) ? 4 * (e0 + 1) : 16 + 4 * (e0 + 1 - 4)] = t1;
                                }
                            }
                            if (orient == 0) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) vc = __builtin_amdgcn_mfma_f64_16x16x4f64(wt[r], w0[(size_t)(4 * r) * RSW + 2 + col16], vc, 0, 0, 0);
#pragma unroll
                                for (int r = 0; r < 4; ++r) vc = __builtin_amdgcn_mfma_f64_16x16x4f64(wt[4 + r], w0[(size_t)(16 + 4 * r) * RSW + 2 + col16], vc, 0, 0, 0);
                            }
                        }
                    }
                    if (rowsum && ch > 0) take_add(ch - 1, prev_rows);
                    if (more) fill_store(ch + 1, s_stage + ((ch + 1) & 1) * CHW * RSW, nfx, nfb);
                    __syncthreads();
                }
                if (rowsum) take_add(nchunk - 1, take_load(nchunk - 1));     // (the next sweep's first barrier orders it before any new partial)
                if (act) {
                    double csum = (csum0 + csum1) * colf;
                    csum += __shfl_xor(csum, 16, 64);
                    csum += __shfl_xor(csum, 32, 64);
                    if (grp == 0) { f_c[col16] = csum; f_cf[col16] = colf; }
                    wave_lds_sync();
                    if (orient == 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) f_V[(4 * r + grp) * 16 + col16] = vc[r] * f_cf[4 * r + grp];      // V[col = 4 r + grp][dim = col16]
                    }
                    wave_lds_sync();
                    fold_tile(orient, il_r, il_c);
                    wave_lds_sync();
                }
            }
        }
        if (rowsum) {
            // the row-side terms (sum_i r_i u_i u_i^T, sum_i r_i nu_ix / l_ax^2) from the accumulated row sums: what the second
            // orientation's fold did with its column sums, 16 side-a points at a time
            __syncthreads();
            for (int ct = wave; ct < NT16; ct += NW) {
                load_columns(ct, s_ila);
                if (lane < 16) {
                    const int jj = ct * 16 + lane;
                    f_c[lane] = jj < N ? s_r[jj] : 0.0;
                }
                wave_lds_sync();
                fold_tile(1, s_ilb, s_ila);
                wave_lds_sync();
            }
        }
    } else
    for (int orient = 0; orient < (diag ? 1 : 2); ++orient) {
        // orient 0: rows are side a (u), columns side b (w), c_ij = u_i . (Z w_j); orient 1: rows side b, columns side a, Z^T
        const int rs = orient ? b : a, cs_ = orient ? a : b;
        const double* il_r = orient ? s_ilb : s_ila;
        const double* il_c = orient ? s_ila : s_ilb;
        const double* k_r = orient ? s_kb : s_ka;
        const double* k_c = orient ? s_ka : s_kb;
        const double* Zu = s_Z + (orient ? 256 : 0);
        const double* beta_r = p.beta + (size_t)rs * N;
        const double* beta_c = p.beta + (size_t)cs_ * N;
        for (int ct = wave; ct < NT16; ct += NW) {
            const int j = ct * 16 + col16;
            const bool jin = j < N;
            const int jc = jin ? j : N - 1;
            // the column's vector x_j (16 values) -> fold scratch, and h_j = Z x_j as the B operand of the c tiles
            for (int k = lane; k < 256; k += 64) {
                const int cc = k >> 4, d = k & 15;
                int jj = ct * 16 + cc;
                jj = jj < N ? jj : N - 1;
                f_w[k] = (d < D) ? (p.Xt[(size_t)d * N + jj] - s_m[d]) * il_c[d] : 0.0;
            }
            for (int k = lane; k < 16 * NX; k += 64) {
                const int cc = k / NX, x = k - cc * NX;
                int jj = ct * 16 + cc;
                jj = jj < N ? jj : N - 1;
                f_x[cc * FX + x] = p.Xt[(size_t)(D + x) * N + jj] - s_m[D + x];
            }
            wave_lds_sync();
            double hB[4];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int d = 4 * qd + grp;
                double h = 0.0;
#pragma unroll
                for (int e = 0; e < 16; ++e) h = fma(Zu[d * 16 + e], f_w[col16 * 16 + e], h);
                hB[qd] = h;
            }
            const double kcj = k_c[jc];
            const double bcj = jin ? beta_c[jc] : 0.0;
            double csum = 0.0;
            mfma_d4 vc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
            for (int rt = 0; rt < NT16; ++rt) {
                // A operand of the c tile: u_row[dim 4 qd + grp], row = rt * 16 + col16
                const int ir = rt * 16 + col16;
                const int irc = ir < N ? ir : N - 1;
                double aA[4];
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int d = 4 * qd + grp;
                    aA[qd] = (d < D) ? (p.Xt[(size_t)d * N + irc] - s_m[d]) * il_r[d] : 0.0;
                }
                mfma_d4 cc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) cc = __builtin_amdgcn_mfma_f64_16x16x4f64(aA[qd], hB[qd], cc, 0, 0, 0);
                double ev[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = rt * 16 + 4 * r + grp;               // the row of accumulator register r
                    const bool iin = i < N;
                    const int ic = iin ? i : N - 1;
                    double wgt;
                    if (diag) wgt = beta_r[ic] * bcj - p.iK[((size_t)a * N + ic) * N + jc];
                    else wgt = beta_r[ic] * bcj;
                    wgt = (iin && jin) ? wgt : 0.0;
                    ev[r] = wgt * fast_exp(k_r[ic] + kcj + cc[r], s_exptab);
                }
                csum += (ev[0] + ev[1]) + (ev[2] + ev[3]);
                if (orient == 0) {
                    // V[col][dim] += sum_rows E[row][col] u_row[dim]: A = the E tile as it lies (register r = k-slice r), B = u rows
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = rt * 16 + 4 * r + grp;
                        const int ic = i < N ? i : N - 1;
                        const double ub = (col16 < D) ? (p.Xt[(size_t)col16 * N + ic] - s_m[col16]) * il_r[col16] : 0.0;
                        vc = __builtin_amdgcn_mfma_f64_16x16x4f64(ev[r], ub, vc, 0, 0, 0);
                    }
                }
            }
            // column sums over the four lane groups that share a column
            csum += __shfl_xor(csum, 16, 64);
            csum += __shfl_xor(csum, 32, 64);
            if (grp == 0) f_c[col16] = csum;
            if (orient == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) f_V[(4 * r + grp) * 16 + col16] = vc[r];      // V[col = 4 r + grp][dim = col16]
            }
            wave_lds_sync();
            // fold the 16 columns into the moments (lanes own entries)
            if (orient == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int e = lane * 4 + k, m_ = e >> 4, n_ = e & 15;
                    double s = 0.0;
                    _Pragma("unroll 2") for (int cc = 0; cc < 16; ++cc) {
                        const double cj = f_c[cc], wm = f_w[cc * 16 + m_], wn = f_w[cc * 16 + n_];
                        s = fma(cj * wm, wn, s);
                        s = fma(f_V[cc * 16 + m_], wn, s);
                        s = fma(wm, f_V[cc * 16 + n_], s);
                        if (diag) s = fma(cj * wm, wn, s);            // symmetric pair: the row-side term equals the column-side one
                    }
                    tP2[k] += s;
                }
                if (lane < 16) {
                    double s = 0.0;
                    _Pragma("unroll 2") for (int cc = 0; cc < 16; ++cc) s += f_V[cc * 16 + lane] + f_c[cc] * f_w[cc * 16 + lane];
                    tP1 += s;
                }
                if (lane < NX) {
                    double s = 0.0;
                    _Pragma("unroll 2") for (int cc = 0; cc < 16; ++cc) s = fma(f_c[cc], f_x[cc * FX + lane], s);
                    tPe += s * (il_c[D + lane] + (diag ? il_r[D + lane] : 0.0));
                }
                if (lane == 0) {
                    double s = 0.0;
                    _Pragma("unroll 2") for (int cc = 0; cc < 16; ++cc) s += f_c[cc];
                    tW += s;
                }
            } else {
                // other orientation: its column sums are the row sums r_i of the pair; columns here are side-a points (u)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int e = lane * 4 + k, m_ = e >> 4, n_ = e & 15;
                    double s = 0.0;
                    _Pragma("unroll 2") for (int cc = 0; cc < 16; ++cc) s = fma(f_c[cc] * f_w[cc * 16 + m_], f_w[cc * 16 + n_], s);
                    tP2[k] += s;
                }
                if (lane < NX) {
                    double s = 0.0;
                    _Pragma("unroll 2") for (int cc = 0; cc < 16; ++cc) s = fma(f_c[cc], f_x[cc * FX + lane], s);
                    tPe += s * il_c[D + lane];
                }
            }
            wave_lds_sync();
        }
    }
    // per-wavefront totals -> LDS -> fixed-order sum -> HBM
    __syncthreads();                                   // s_tot shares its space with the stage
    {
        double* tw = s_tot + wave * NSP;
        for (int k = lane; k < NSP; k += 64) tw[k] = 0.0;
        wave_lds_sync();
        if (lane == 0) tw[0] = tW;
        if (lane < 16 && lane < D) tw[1 + lane] = tP1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = lane * 4 + k, m_ = e >> 4, n_ = e & 15;
            if (m_ < D && n_ < D) tw[1 + D + m_ * D + n_] = tP2[k];
        }
        if (lane < NX) tw[1 + D + D * D + lane] = tPe;
    }
    __syncthreads();
    double* out = p.mom + (((size_t)c * H + t) * (D * (D + 1) / 2) + q) * NSP;
    for (int k = tid; k < NSP; k += NT) {
        double s = 0.0;
        for (int w = 0; w < NW; ++w) s += s_tot[w * NSP + k];
        out[k] = s;
    }
}

// ------------------------------------------------------------------------------------------
// Dense D x D helpers for ONE wavefront on matrices in LDS (row-major, stride D): lanes own entries.
// C = op(A) op(B); tA / tB: use the transpose.  Call with the whole wavefront; ends with a wave-level LDS sync.
__device__ inline void wave_matmul(double* C, const double* A, const double* B, int D, bool tA, bool tB, int lane) {
    for (int idx = lane; idx < D * D; idx += 64) {
        const int i = idx / D, j = idx - i * D;
        double s = 0.0;
        for (int k = 0; k < D; ++k) s = fma(tA ? A[k * D + i] : A[i * D + k], tB ? B[j * D + k] : B[k * D + j], s);
        C[idx] = s;
    }
    wave_lds_sync();
}

struct WideSweepLayout {
    int ils2, var, cost, m, mu, Sig, M, Sb, Vb, Mb, Sbar, mbar, mubar, gmu, gSig, gu, ctmp, aug, Ai, small, G2, chunk, red, pw, pacc, total;
};

__host__ __device__ inline WideSweepLayout make_wide_sweep_layout(int D, int A, int E) {
    WideSweepLayout L;
    const int DD = D * D, n = D + A, NX = E - D;
    int o = 0;
    auto take = [&](int& f, int sz) { f = o; o += rnd2(sz); };
    take(L.ils2, D * E); take(L.var, D); take(L.cost, n + n * n + DD + 2 * D);
    take(L.m, E); take(L.mu, D); take(L.Sig, DD); take(L.M, D);
    take(L.Sb, DD); take(L.Vb, DD); take(L.Mb, D); take(L.Sbar, DD); take(L.mbar, E); take(L.mubar, D);
    take(L.gmu, D); take(L.gSig, DD); take(L.gu, A > 0 ? A : 1); take(L.ctmp, 2 * n * n + 3 * n);
    take(L.aug, 2 * DD); take(L.Ai, DD);
    take(L.small, 8 * D + 8);                 // s1, y, vb, s1b, G1, Ai G1 | scalars
    const int phase0 = o;
    take(L.G2, 4 * DD + 4 * (D + NX + 1));    // four partial copies of G2 | G1, Ge, s0
    take(L.chunk, 256 * (D + NX + 2));        // per point of the chunk: nu (D), nu_x (NX), lb, om
    take(L.red, 16 * (D + 1));
    const int phase0_end = o;
    o = phase0;                               // the pair phase reuses the point-pass buffers (the phases are barrier-separated)
    take(L.pw, kWidePairWaves * (2 * DD + 4 * DD));       // per pair wavefront: aug (2 DD) | Ri | Z | T1 | T2
    if (o < phase0_end) o = phase0_end;
    take(L.pacc, kWidePairWaves * (DD + E));              // per pair wavefront: its Sigma_bar and m_bar contributions
    L.total = o;
    return L;
}

template <int DP>
__global__ __launch_bounds__(kWideSweepThreads) void wide_adjoint_sweep_kernel(const WideArgs p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int NT = kWideSweepThreads;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = blockIdx.x;
    const int N = p.N, D = p.D, A = p.A, E = p.E, H = p.H;
    const int NX = E - D, DD = D * D, n = D + A, LD = 2 * D, P = D * (D + 1) / 2, NSP = p.NSP;
    const WideSweepLayout L = make_wide_sweep_layout(D, A, E);
    double* c_ils2 = smem + L.ils2; double* c_var = smem + L.var; double* c_cost = smem + L.cost;
    double* s_m = smem + L.m; double* s_mu = smem + L.mu; double* s_Sig = smem + L.Sig; double* s_M = smem + L.M;
    double* s_Sb = smem + L.Sb; double* s_Vb = smem + L.Vb; double* s_Mb = smem + L.Mb; double* s_Sbar = smem + L.Sbar;
    double* s_mbar = smem + L.mbar; double* s_mubar = smem + L.mubar;
    double* s_gmu = smem + L.gmu; double* s_gSig = smem + L.gSig; double* s_gu = smem + L.gu; double* s_ctmp = smem + L.ctmp;
    double* s_aug = smem + L.aug; double* s_Ai = smem + L.Ai; double* s_small = smem + L.small; double* s_G2 = smem + L.G2;
    double* s_chunk = smem + L.chunk; double* s_pw = smem + L.pw; double* s_pacc = smem + L.pacc;
    double* s_s1 = s_small; double* s_y = s_small + D; double* s_vb = s_small + 2 * D; double* s_s1b = s_small + 3 * D;
    double* s_G1 = s_small + 4 * D; double* s_AiG1 = s_small + 5 * D; double* s_sc = s_small + 8 * D;   // scalars: c, s0, cb, s0b
    const int CW = D + NX + 2;                                    // words per point of the chunk

    for (int i = tid; i < D * E; i += NT) c_ils2[i] = p.ils2[i];
    for (int i = tid; i < D; i += NT) c_var[i] = p.var[i];
    for (int i = tid; i < n + n * n + DD + 2 * D; i += NT) c_cost[i] = p.cost[i];
    __syncthreads();
    const double* target = c_cost; const double* Wm = c_cost + n; const double* WT = Wm + n * n;
    const double* smin = WT + DD; const double* smax = smin + D;
    const double inv_n = 1.0 / (double)(H + 1);

    // terminal cost adjoint: (mu_bar, Sigma_bar) at t = H  (setpoint_distance_reward_mapper.py:135-141, gp_mpc_controller.py:270-276)
    if (wave == 0) {
        const double* muH = p.mu + ((size_t)c * (H + 1) + H) * D;
        const double* SgH = p.Sig + ((size_t)c * (H + 1) + H) * DD;
        for (int i = lane; i < D; i += 64) s_mu[i] = muH[i];
        for (int i = lane; i < DD; i += 64) s_Sig[i] = SgH[i];
        wave_lds_sync();
        const double cvH = p.cv[(size_t)c * (H + 1) + H];
        cost_adjoint_wave(lane, D, 0, true, s_mu, s_Sig, s_mu, target, WT, smin, smax, false, inv_n,
                          -p.kappa / (2.0 * sqrt(cvH)) * inv_n, s_ctmp, s_gmu, s_gSig, s_gu);
        wave_lds_sync();
        for (int i = lane; i < D; i += 64) s_mubar[i] = s_gmu[i];
        for (int i = lane; i < DD; i += 64) { const int r = i / D, cc = i - r * D; s_Sbar[i] = 0.5 * (s_gSig[i] + s_gSig[cc * D + r]); }
    }
    __syncthreads();

    for (int t = H - 1; t >= 0; --t) {
        // ---- state of step t, seeds of the adjoints (the checker's backward_step) ------------------------------------
        {
            const double* mut = p.mu + ((size_t)c * (H + 1) + t) * D;
            const double* Sgt = p.Sig + ((size_t)c * (H + 1) + t) * DD;
            for (int i = tid; i < D; i += NT) { s_mu[i] = mut[i]; s_M[i] = mut[D + i] - mut[i]; }       // M = mu_{t+1} - mu_t
            for (int i = tid; i < DD; i += NT) s_Sig[i] = Sgt[i];
            for (int e = tid; e < E; e += NT) {
                double v;
                if (e < D) v = mut[e];
                else if (e < D + A) v = p.actions[((size_t)c * H + t) * A + (e - D)];
                else v = p.time0 + (double)t;
                s_m[e] = v;
            }
            for (int i = tid; i < DD; i += NT) { const int r = i / D, cc = i - r * D; s_Sb[i] = 0.5 * (s_Sbar[i] + s_Sbar[cc * D + r]); }
        }
        __syncthreads();
        for (int i = tid; i < DD; i += NT) {
            const int r = i / D, cc = i - r * D;
            double v = 0.0;
            for (int k = 0; k < D; ++k) v = fma(s_Sig[r * D + k], 2.0 * s_Sb[k * D + cc], v);       // Vb = Sigma (2 Sb)
            s_Vb[i] = v;
        }
        for (int i = tid; i < D; i += NT) {
            double v = s_mubar[i];
            for (int k = 0; k < D; ++k) v = fma(-2.0 * s_Sb[i * D + k], s_M[k], v);                   // Mb = mu_bar' - 2 Sb M
            s_Mb[i] = v;
        }
        __syncthreads();
        for (int i = tid; i < DD; i += NT) s_Sbar[i] = s_Sb[i];                                       // Sigma_bar starts as Sb
        for (int e = tid; e < E; e += NT) s_mbar[e] = e < D ? s_mubar[e] : 0.0;
        __syncthreads();

        // ---- mean part, one output a at a time -------------------------------------------------------------------
        for (int a = 0; a < D; ++a) {
            const double* il = c_ils2 + a * E;
            if (wave == 0) {
                double prodil = 1.0;
                for (int i = 0; i < D; ++i) prodil *= il[i];
                for (int idx = lane; idx < DD; idx += 64) {
                    const int i = idx / D, j = idx - i * D;
                    s_aug[i * LD + j] = s_Sig[idx] + (i == j ? 1.0 / il[i] : 0.0);
                    s_aug[i * LD + D + j] = (i == j ? 1.0 : 0.0);
                }
                wave_lds_sync();
                const double detA = wave_gauss_solve(s_aug, D, D, LD, lane);
                for (int idx = lane; idx < DD; idx += 64) { const int i = idx / D, j = idx - i * D; s_Ai[idx] = s_aug[i * LD + D + j]; }
                if (lane == 0) s_sc[0] = c_var[a] / sqrt(detA * prodil);
            }
            __syncthreads();
            // two passes over the points in chunks of 256: pass 0 -> s0, s1; pass 1 -> moments of om_i = -lb_i (s0b + nu_i . s1b) / 2
            for (int pass = 0; pass < 2; ++pass) {
                for (int k = tid; k < 4 * DD + 4 * (D + NX + 1); k += NT) s_G2[k] = 0.0;
                __syncthreads();
                double g2acc = 0.0, g1acc = 0.0;       // thread (g = tid >> 8, e = tid & 255): G2 entry e over the points p = g mod (NT / 256)
                for (int c0 = 0; c0 < N; c0 += 256) {
                    if (tid < 256) {
                        const int pt = c0 + tid;
                        double* rec = s_chunk + tid * CW;
                        if (pt < N) {
                            double qv = 0.0;
                            for (int d = 0; d < D; ++d) rec[d] = p.Xt[(size_t)d * N + pt] - s_m[d];
                            for (int x = 0; x < NX; ++x) {
                                const double v = p.Xt[(size_t)(D + x) * N + pt] - s_m[D + x];
                                rec[D + x] = v;
                                qv = fma(v * v, il[D + x], qv);
                            }
                            for (int i = 0; i < D; ++i) {
                                double r = 0.0;
                                for (int j = 0; j < D; ++j) r = fma(s_Ai[i * D + j], rec[j], r);
                                qv = fma(rec[i], r, qv);
                            }
                            const double lb = exp(-0.5 * qv) * p.beta[(size_t)a * N + pt];
                            rec[D + NX] = lb;
                            double om = 0.0;
                            if (pass == 1) {
                                double dot = s_sc[3];
                                for (int d = 0; d < D; ++d) dot = fma(rec[d], s_s1b[d], dot);
                                om = -0.5 * lb * dot;
                            }
                            rec[D + NX + 1] = om;
                        } else {
                            for (int k = 0; k < CW; ++k) rec[k] = 0.0;
                        }
                    }
                    __syncthreads();
                    {
                        const int g = tid >> 8, e = tid & 255;
                        const int wsel = pass == 0 ? D + NX : D + NX + 1;            // weight: lb (pass 0) or om (pass 1)
                        if (pass == 1 && e < DD) {
                            const int d1 = e / D, d2 = e - d1 * D;
                            for (int pp = g; pp < 256; pp += NT / 256) {
                                const double* rec = s_chunk + pp * CW;
                                g2acc = fma(rec[wsel] * rec[d1], rec[d2], g2acc);
                            }
                        }
                        // first moments: entries 0 .. D-1 nu_d, D .. D+NX-1 nu_x, D+NX: the weight itself
                        if (e < D + NX + 1) {
                            for (int pp = g; pp < 256; pp += NT / 256) {
                                const double* rec = s_chunk + pp * CW;
                                g1acc = fma(rec[wsel], e < D + NX ? rec[e] : 1.0, g1acc);
                            }
                        }
                    }
                    __syncthreads();
                }
                {
                    const int g = tid >> 8, e = tid & 255;
                    if (e < DD) s_G2[g * DD + e] = g2acc;
                    if (e < D + NX + 1) s_G2[4 * DD + g * (D + NX + 1) + e] = g1acc;
                }
                __syncthreads();
                if (pass == 0) {
                    // s0, s1 -> y = Ai s1, cb, s0b, s1b   (wave 0)
                    if (wave == 0) {
                        const double* f1 = s_G2 + 4 * DD;
                        const int F = D + NX + 1;
                        for (int d = lane; d < D; d += 64) s_s1[d] = (f1[d] + f1[F + d]) + (f1[2 * F + d] + f1[3 * F + d]);
                        if (lane == 0) s_sc[1] = (f1[D + NX] + f1[F + D + NX]) + (f1[2 * F + D + NX] + f1[3 * F + D + NX]);      // s0
                        for (int d = lane; d < D; d += 64) s_vb[d] = s_Vb[d * D + a];
                        wave_lds_sync();
                        for (int d = lane; d < D; d += 64) {
                            double y = 0.0, s1b = 0.0;
                            for (int k = 0; k < D; ++k) { y = fma(s_Ai[d * D + k], s_s1[k], y); s1b = fma(s_Ai[d * D + k], s_vb[k], s1b); }
                            s_y[d] = y;
                            s_s1b[d] = s_sc[0] * s1b;
                        }
                        wave_lds_sync();
                        if (lane == 0) {
                            double vy = 0.0;
                            for (int k = 0; k < D; ++k) vy = fma(s_vb[k], s_y[k], vy);
                            s_sc[2] = s_Mb[a] * s_sc[1] + vy;                                  // cb
                            s_sc[3] = s_Mb[a] * s_sc[0];                                       // s0b
                        }
                    }
                    __syncthreads();
                }
            }
            // ---- adjoints of output a:  Aib = sym(c vb s1^T + G2), m_bar, Ab, Sigma_bar ---------------------------------
            {
                const double* f1 = s_G2 + 4 * DD;
                const int F = D + NX + 1;
                for (int i = tid; i < DD; i += NT) s_aug[i] = (s_G2[i] + s_G2[DD + i]) + (s_G2[2 * DD + i] + s_G2[3 * DD + i]);      // G2
                for (int d = tid; d < D; d += NT) s_G1[d] = (f1[d] + f1[F + d]) + (f1[2 * F + d] + f1[3 * F + d]);
                for (int x = tid; x < NX; x += NT) {
                    const double ge = (f1[D + x] + f1[F + D + x]) + (f1[2 * F + D + x] + f1[3 * F + D + x]);
                    s_mbar[D + x] -= 2.0 * il[D + x] * ge;
                }
            }
            __syncthreads();
            for (int i = tid; i < DD; i += NT) {
                const int r = i / D, cc = i - r * D;
                const double v1 = s_sc[0] * s_vb[r] * s_s1[cc] + s_aug[i], v2 = s_sc[0] * s_vb[cc] * s_s1[r] + s_aug[cc * D + r];
                s_aug[DD + i] = 0.5 * (v1 + v2);                                                    // Aib (symmetrised)
            }
            for (int d = tid; d < D; d += NT) {
                double g = 0.0;
                for (int k = 0; k < D; ++k) g = fma(s_Ai[d * D + k], s_G1[k], g);
                s_AiG1[d] = g;
            }
            __syncthreads();
            for (int d = tid; d < D; d += NT) s_mbar[d] -= s_sc[1] * s_s1b[d] + 2.0 * s_AiG1[d];
            // T = Ai Aib (into the first half of aug), then Ab = -T Ai - cb c Ai / 2
            for (int i = tid; i < DD; i += NT) {
                const int r = i / D, cc = i - r * D;
                double v = 0.0;
                for (int k = 0; k < D; ++k) v = fma(s_Ai[r * D + k], s_aug[DD + k * D + cc], v);
                s_aug[i] = v;
            }
            __syncthreads();
            for (int i = tid; i < DD; i += NT) {
                const int r = i / D, cc = i - r * D;
                double v = 0.0;
                for (int k = 0; k < D; ++k) v = fma(s_aug[r * D + k], s_Ai[k * D + cc], v);
                // Sigma_bar += Ab + (Cb V^T) contribution of output a: Cb[r][a] V[cc][a], V[:, a] = c y
                s_Sbar[i] += -v - 0.5 * s_sc[2] * s_sc[0] * s_Ai[i] + 2.0 * s_Sb[r * D + a] * (s_sc[0] * s_y[cc]);
            }
            __syncthreads();
        }
        // ---- pairs: kWidePairWaves wavefronts, each with its own scratch and its own accumulators --------------------------
        if (wave < kWidePairWaves) {
            double* aug = s_pw + wave * (6 * DD);
            double* Ri = aug + 2 * DD; double* Z = Ri + DD; double* T1 = Z + DD; double* T2 = T1 + DD;
            double* accS = s_pacc + wave * (DD + E);
            double* accm = accS + DD;
            for (int i = lane; i < DD + E; i += 64) accS[i] = 0.0;
            wave_lds_sync();
            int q = 0;
            for (int a = 0; a < D; ++a)
                for (int b = a; b < D; ++b, ++q) {
                    if ((q % kWidePairWaves) != wave) continue;
                    const double* ila = c_ils2 + a * E; const double* ilb = c_ils2 + b * E;
                    for (int idx = lane; idx < DD; idx += 64) {
                        const int i = idx / D, j = idx - i * D;
                        aug[i * LD + j] = s_Sig[idx] * (ila[j] + ilb[j]) + (i == j ? 1.0 : 0.0);
                        aug[i * LD + D + j] = (i == j ? 1.0 : 0.0);
                    }
                    wave_lds_sync();
                    const double detR = wave_gauss_solve(aug, D, D, LD, lane);
                    for (int idx = lane; idx < DD; idx += 64) { const int i = idx / D, j = idx - i * D; Ri[idx] = aug[i * LD + D + j]; }
                    wave_lds_sync();
                    wave_matmul(Z, Ri, s_Sig, D, false, false, lane);                               // Z = R^-1 Sigma
                    const double rdet = 1.0 / sqrt(detR);
                    const double* mom = p.mom + (((size_t)c * H + t) * P + q) * NSP;
                    const double W_ = mom[0];
                    const double* P1 = mom + 1; const double* P2 = mom + 1 + D; const double* Pe = mom + 1 + D + DD;
                    const double sb = (a == b) ? s_Sb[a * D + b] : 2.0 * s_Sb[a * D + b];
                    const double Wb = sb * rdet;
                    // m_bar[:D] += Wb (P1 - dab . (Z P1));  m_bar[D:] += Wb Pe
                    for (int d = lane; d < D; d += 64) {
                        double zp = 0.0;
                        for (int k = 0; k < D; ++k) zp = fma(Z[d * D + k], P1[k], zp);
                        accm[d] += Wb * (P1[d] - (ila[d] + ilb[d]) * zp);
                    }
                    for (int x = lane; x < NX; x += 64) accm[D + x] += Wb * Pe[x];
                    // Zb = Wb P2 / 2;  T1 = Ri^T Zb;  Sigma_bar += T1;  Rb = -sb W rdet Ri^T / 2 - T1 Z^T;  Sigma_bar += Rb . dab (columns)
                    for (int idx = lane; idx < DD; idx += 64) T2[idx] = 0.5 * Wb * P2[idx];
                    wave_lds_sync();
                    wave_matmul(T1, Ri, T2, D, true, false, lane);                                  // T1 = Ri^T Zb
                    wave_matmul(T2, T1, Z, D, false, true, lane);                                   // T2 = T1 Z^T
                    for (int idx = lane; idx < DD; idx += 64) {
                        const int i = idx / D, j = idx - i * D;
                        const double Rb = -0.5 * sb * W_ * rdet * Ri[j * D + i] - T2[idx];
                        accS[idx] += T1[idx] + Rb * (ila[j] + ilb[j]);
                    }
                    wave_lds_sync();
                }
        }
        __syncthreads();
        for (int i = tid; i < DD; i += NT) {
            double v = s_Sbar[i];
            for (int w = 0; w < kWidePairWaves; ++w) v += s_pacc[w * (DD + E) + i];
            s_Sbar[i] = v;
        }
        for (int e = tid; e < E; e += NT) {
            double v = s_mbar[e];
            for (int w = 0; w < kWidePairWaves; ++w) v += s_pacc[w * (DD + E) + DD + e];
            s_mbar[e] = v;
        }
        __syncthreads();
        // ---- stage cost of step t, hand-over to step t - 1 -------------------------------------------------------------
        if (wave == 0) {
            const double cvt = p.cv[(size_t)c * (H + 1) + t];
            const double* act = p.actions + ((size_t)c * H + t) * A;
            cost_adjoint_wave(lane, D, A, false, s_mu, s_Sig, act, target, Wm, smin, smax, p.use_constraints != 0, inv_n,
                              -p.kappa / (2.0 * sqrt(cvt)) * inv_n, s_ctmp, s_gmu, s_gSig, s_gu);
            wave_lds_sync();
            for (int k = lane; k < A; k += 64) p.grad[((size_t)c * H + t) * A + k] = s_mbar[D + k] + s_gu[k];
            for (int i = lane; i < D; i += 64) s_mubar[i] = s_mbar[i] + s_gmu[i];
        }
        __syncthreads();
        for (int i = tid; i < DD; i += NT) {
            const int r = i / D, cc = i - r * D;
            // Sigma_bar symmetrised, plus the symmetric part of the stage-cost partial
            s_Vb[i] = 0.5 * (s_Sbar[i] + s_Sbar[cc * D + r]) + 0.5 * (s_gSig[i] + s_gSig[cc * D + r]);
        }
        __syncthreads();
        for (int i = tid; i < DD; i += NT) s_Sbar[i] = s_Vb[i];
        __syncthreads();
    }
}

}  // namespace gpmpc_hip
