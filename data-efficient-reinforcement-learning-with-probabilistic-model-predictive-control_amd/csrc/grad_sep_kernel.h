// grad_sep_kernel.h -- moments of the OFF-DIAGONAL output pairs for the analytic gradient, in separable form on the fp64
// matrix cores (gfx950 / MI355X; D <= 4).
//
// The reverse sweep needs, per output pair (a, b) and horizon step, the moments of p_ij = u_i + w_j under the pairwise
// weights E_ij (grad_kernels.h):  W = sum E,  P1 = sum E p,  P2 = sum E p p^T,  Pe = sum E (nu_ix / l_ax^2 + nu_jx / l_bx^2).
// pair_moments_kernel evaluates them element by element -- N^2 elements x ~32 instructions per pair -- and the three
// (config 2) or six (config 4) off-diagonal pairs are two thirds / three quarters of that work.  For a != b the weights
// factor,  E_ij = ra_i rb_j exp(g_i . w_j),  and with the Taylor polynomial of exp(g . w) the forward kernels already use
// (degree K from the data range, truncation below fp64 rounding)  exp(g . w) = sum_|alpha|<=K g^alpha w^alpha / alpha!,
// so every moment is a sum over monomials of (row-side moment) x (column-side moment):
//     W  = sum_alpha c_alpha G[1][alpha] H[1][alpha]                                   c_alpha = 1 / alpha!
//     P1 = sum_alpha c_alpha (G[u][alpha] H[1][alpha] + G[1][alpha] H[w][alpha])
//     P2 = sum_alpha c_alpha (G[u u^T] H[1] + G[u] H[w]^T + H[w] G[u]^T + G[1] H[w w^T])
//     Pe = sum_alpha c_alpha (G[nu_x] H[1] / l_ax^2 + G[1] H[nu_x] / l_bx^2)
//   G[f][alpha] = sum_i ra_i f(i) g_i^alpha  over the weightings f in {1, u_d, u_d u_e, nu_x},  H likewise with rb_j, w_j.
// That is O(N (#weightings x #monomials)) per pair and side instead of O(N^2), and G = F^T Phi is a genuine matrix product
// with the POINTS as the inner dimension: 16 weightings x 16 monomials per v_mfma_f64_16x16x4_f64, four points per
// instruction, no cross-lane reduction anywhere (the element-wise form of this idea needed 4 x 16 accumulators per lane
// and spilled: DESIGN.md section 4.4).  A wavefront owns one (pair, side): per 32 points it writes, per point, the
// weightings wt_i f(v_i) themselves and two small tables of products of the monomial variables (x_0^i x_1^j | x_2^i x_3^j) to
// LDS (two lanes per point), then 8 k-steps of MFMAs whose A operand is ONE table read and whose B operand is the product of
// two (sep_grad_point_words).
//
// Pairs whose degree is outside the table (direct-exp form, K beyond kSepGradBlocks x 16 monomials) are left to the
// element-wise kernels: `done[(c, t, pair)]` says which pairs this kernel wrote.
#pragma once
#include "rollout_stream_kernel.h"
#include "grad_kernels.h"

namespace gpmpc_hip {

// wavefronts per (candidate, step): two pairs x two sides at a time (six at D = 3 -- all tasks in one round -- measured slower:
// 1.11 vs 0.75 ms at config 2, 77 KB of LDS per workgroup); twelve regions of moment matrices at D = 4 would not fit the LDS
__host__ __device__ constexpr int sep_grad_waves(int DP) { return DP == 2 ? 2 : 4; }
constexpr int kSepGradBlocks = 5;            // monomial blocks of 16 per (pair, side): up to 80 monomials

struct SepGradArgs {
    const double* Xt;       // (E, N)
    const double* beta;     // (D, N)
    const double* ils2;     // (D, E)
    const double* logvar;   // (D)
    const double* xrange;   // (2, E)
    const double* actions;  // (B, H, A)
    const double* mu;       // (B, H + 1, D)
    const double* Sig;      // (B, H + 1, D, D)
    const int* mono_exp;    // (CM, 4) exponents, graded order (rollout.hip ensure_monomials)
    const double* mono_w;   // (CM) 1 / alpha!
    double* mom;            // (B, H, P, NSP)   [W | P1 (DP) | P2 upper triangle | Pe (NXP)]
    int* done;              // (B, H, P) 1 = the pair's moments were written here
    int mono_cum[16];       // monomials of degree <= k
    int N, D, A, E, H, B;
    int include_time;
    double time0;
    int NSP, NXP;
    int kmax;               // highest degree this kernel takes (mono_cum[kmax] <= 16 kSepGradBlocks)
    int force_path;
    int keep_diag_flags;    // 1: the diagonal pairs' flags were written by the fused forward: leave them
    int PS;                 // doubles per point in the LDS tables
    int wave_words;         // doubles of LDS per wavefront
};

// Per-point table of a 32-point chunk (round 4): the FW = 16 NA + NE weightings wt f(v) themselves | the products x_a^i x_b^j of
// the first two monomial variables (i + j <= K, triangular) | those of the last two (one variable: its powers).  A k-step then reads ONE value per weighting block and TWO per monomial block where
// the first version read 3 and D of them (wt, two factors; one power per variable) -- 17 individually addressed reads per
// lane and 4-point step at config 4, two thirds of whose LDS cycles were bank conflicts (profiles/r03_c4_gradient_pmc.txt:
// 3.29e9 conflict cycles against 4.90e9 active), next to ~25 multiplies that formed the operands.
// Stride: an ODD number of doubles.  The table pass WRITES with lanes = points (every lane the same slot of its own point): at
// an even stride the 16 lanes of a store group share banks -- at 48 doubles (= 0 mod 32 dwords, chosen first because it puts
// the two points a 32-lane half READS 32 banks apart) all of them hit ONE bank, 64 LDS cycles per store instead of 4, and the
// pass got slower than its predecessor (87 / 83 ms at config 4 against 66: profiles/r04e_sepv2a_*, r04f_sepv2b_*).  At an odd
// stride the stores are conflict-free and the reads nearly so (the 16 lanes of a point read 16 consecutive doubles or entries
// of a table of <= 15 doubles; two points 2 x odd dwords apart overlap in at most one bank pair).
// first version (kept for D <= 3, see sep_grad_version): wt | v = (1, u_0 .., nu_x ..) | powers x_d^e, e = 0 .. K | one zero; odd stride
__host__ __device__ inline int sep_grad_point_words_v1(int D, int NX, int K) {
    const int n = 1 + (1 + D + NX) + D * (K + 1) + 1;
    return n | 1;
}
// Which table form a state dimension takes.  Version 2 (weightings + pair-product tables, 32-point chunks) wins where the
// k-steps dominate: D = 4 (config 4: 66.4 -> 52.6 ms).  At D <= 3 the tables of 64 points no longer fit the wavefront's LDS
// region with it, and with 32-point chunks the per-chunk table pass -- 7 instead of 4 at N = 200 -- outweighs the cheaper
// k-steps of the one or two monomial blocks these shapes have (config 2: 0.45 -> 0.56 ms, config 3 +2 %:
// profiles/r04f_sepv2d_*): they keep version 1 (per-variable power tables, operands assembled in the k-step).
__host__ __device__ constexpr int sep_grad_version(int DP) { return DP >= 4 ? 2 : 1; }

__host__ __device__ inline int sep_grad_tri(int K1, int nv) { return nv == 2 ? K1 * (K1 + 1) / 2 : K1; }
__host__ __device__ inline int sep_grad_nv0(int D) { return D >= 3 ? 2 : 1; }        // variables of the first group
__host__ __device__ inline int sep_grad_point_words(int D, int FW, int K) {
    const int nv0 = sep_grad_nv0(D), nv1 = D - nv0;
    const int n = FW + sep_grad_tri(K + 1, nv0) + sep_grad_tri(K + 1, nv1);
    return n | 1;
}
// points per chunk: 64 (lane = point: weightings and both factor tables) while 64 tables fit the wavefront's LDS region (D <= 3:
// <= 31 doubles per point); 32 at D = 4 (49 doubles per point), two lanes per point sharing the work by variable group
__host__ __device__ constexpr int sep_grad_chunk(int DP) { return DP <= 3 ? 64 : 32; }

// ------------------------------------------------------------------------------------------
// NA: blocks of 16 weightings on the matrix cores; NE: weightings 16 NA .. 16 NA + NE - 1 accumulated by plain FMAs instead (a 17th
// weighting -- config 4: 1 + 4 + 10 + 2 -- would otherwise cost a second, 94 % empty, A block: twice the matrix instructions).
template <int DP, int NA, int NE = 0>
__global__ __launch_bounds__(64 * sep_grad_waves(DP), DP <= 3 ? 3 : 2) void sep_grad_moments_kernel(const SepGradArgs p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int NW = sep_grad_waves(DP), NT = 64 * NW, NB = kSepGradBlocks;
    constexpr int NH = DP * (DP + 1) / 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = blockIdx.x, c = blockIdx.y;
    const int N = p.N, D = p.D, A = p.A, E = p.E, H = p.H;
    const int NX = E - D, P = D * (D + 1) / 2, Poff = P - D;
    const int nW = 1 + D + D * (D + 1) / 2 + NX;         // weightings: 1 | u_d | u_d u_e (d <= e) | nu_x
    const int PS = p.PS;
    // LDS: shared small data, then one region per wavefront (point tables during the pass, its moment matrix afterwards)
    double* s_m = smem;                                  // E
    double* s_ils2 = s_m + rnd2(E);                      // D * E
    double* s_Z = s_ils2 + rnd2(D * E);                  // Poff * DP * DP
    double* s_tab = s_Z + rnd2((Poff > 0 ? Poff : 1) * DP * DP);      // exp table
    int* s_K = reinterpret_cast<int*>(s_tab + 64);       // Poff degrees (0: not taken)
    int* s_q = s_K + 8;                                  // Poff: full pair index
    double* s_cw = s_tab + 64 + 8;                       // 16 NB monomial weights
    int* s_me = reinterpret_cast<int*>(s_cw + 16 * NB);  // 16 NB packed exponents
    double* s_wave = s_cw + 16 * NB + 8 * NB + 8;
    double* tabw = s_wave + (size_t)wave * p.wave_words;

    for (int e = tid; e < E; e += NT) {
        double v;
        if (e < D) v = p.mu[((size_t)c * (H + 1) + t) * D + e];
        else if (e < D + A) v = p.actions[((size_t)c * H + t) * A + (e - D)];
        else v = p.time0 + (double)t;
        s_m[e] = v;
    }
    for (int i = tid; i < D * E; i += NT) s_ils2[i] = p.ils2[i];
    for (int i = tid; i < 64; i += NT) s_tab[i] = kExp2Tab[i];
    for (int i = tid; i < 16 * NB; i += NT) {
        const bool in = i < p.mono_cum[p.kmax];
        s_cw[i] = in ? p.mono_w[i] : 0.0;
        s_me[i] = in ? (p.mono_exp[i * 4] | (p.mono_exp[i * 4 + 1] << 8) | (p.mono_exp[i * 4 + 2] << 16) | (p.mono_exp[i * 4 + 3] << 24)) : 0;
    }
    __syncthreads();
    // diagonal pairs are never taken here (their weights do not factor): say so, the flag array is not initialised otherwise
    if (tid >= 64 && tid < 64 + D && !p.keep_diag_flags) p.done[((size_t)c * H + t) * P + (tid - 64) * D - ((tid - 64) * (tid - 65)) / 2] = 0;
    // ---- D x D algebra of the off-diagonal pairs: Z = R^-1 Sigma, Taylor degree (phase P1 of rollout_kernel) -------------
    if (tid < Poff) {
        int a = 0, b = 0, k = tid, q = 0;
        for (int aa = 0; aa < D; ++aa)
            for (int bb = aa; bb < D; ++bb, ++q)
                if (aa != bb) { if (k == 0) { a = aa; b = bb; s_q[tid] = q; } --k; }
        const double* Sg = p.Sig + ((size_t)c * (H + 1) + t) * D * D;
        double m[DP][2 * DP];
#pragma unroll
        for (int i = 0; i < DP; ++i)
#pragma unroll
            for (int j = 0; j < DP; ++j) {
                const bool in = (i < D && j < D);
                const double sg = in ? Sg[i * D + j] : 0.0;
                m[i][j] = sg * (in ? s_ils2[a * E + j] + s_ils2[b * E + j] : 0.0) + (i == j ? 1.0 : 0.0);
                m[i][DP + j] = sg;
            }
        (void)small_solve<DP>(m);
        double cmax = 0.0;
#pragma unroll
        for (int i = 0; i < DP; ++i) {
            const double ui = (i < D) ? fmax(fabs(p.xrange[i] - s_m[i]), fabs(p.xrange[E + i] - s_m[i])) * s_ils2[a * E + i] : 0.0;
#pragma unroll
            for (int j = 0; j < DP; ++j) {
                const double z = (i < D && j < D) ? m[i][DP + j] : 0.0;
                s_Z[(tid * DP + i) * DP + j] = z;
                const double wj = (j < D) ? fmax(fabs(p.xrange[j] - s_m[j]), fabs(p.xrange[E + j] - s_m[j])) * s_ils2[b * E + j] : 0.0;
                cmax = fma(fabs(z) * ui, wj, cmax);
            }
        }
        int K = 0;
        if (p.force_path == 0 && cmax <= kTaylorMaxArg[kMaxTaylor]) {
            K = 1;
            for (int k2 = 1; k2 < kMaxTaylor; ++k2) K += (cmax > kTaylorMaxArg[k2]) ? 1 : 0;
        }
        if (K > p.kmax) K = 0;
        s_K[tid] = K;
        p.done[((size_t)c * H + t) * P + s_q[tid]] = K > 0 ? 1 : 0;
    }
    __syncthreads();

    const int r16 = lane & 15, kq = lane >> 4;
    const int NV = 1 + D + NX;                           // weighting vector (1, u, nu_x) of the first table form
    // per-lane selectors.  A operand: weighting r = 16 ia + (lane & 15) -> two indices into the weighting vector;
    // B operand: monomial n = 16 ib + (lane & 15) -> packed exponents (read per block below)
    int selA[NA][2];
#pragma unroll
    for (int ia = 0; ia < NA; ++ia) {
        const int r = ia * 16 + r16;
        int s1 = 0, s2 = 0;
        if (r >= 1 && r <= D) s1 = r;
        else if (r > D && r < 1 + D + D * (D + 1) / 2) {
            int k = r - 1 - D, d = 0;
            while (k >= D - d) { k -= D - d; ++d; }
            s1 = 1 + d; s2 = 1 + d + k;
        } else if (r >= 1 + D + D * (D + 1) / 2 && r < nW) s1 = 1 + D + (r - 1 - D - D * (D + 1) / 2);
        selA[ia][0] = (r < nW) ? s1 : -1;
        selA[ia][1] = s2;
    }
    // the extra weightings (wave-uniform selectors)
    int selE[NE > 0 ? NE : 1][2];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int r = NA * 16 + e;
        int s1 = 0, s2 = 0;
        if (r >= 1 && r <= D) s1 = r;
        else if (r > D && r < 1 + D + D * (D + 1) / 2) {
            int k = r - 1 - D, d = 0;
            while (k >= D - d) { k -= D - d; ++d; }
            s1 = 1 + d; s2 = 1 + d + k;
        } else if (r >= 1 + D + D * (D + 1) / 2 && r < nW) s1 = 1 + D + (r - 1 - D - D * (D + 1) / 2);
        selE[e][0] = (r < nW) ? s1 : -1;
        selE[e][1] = s2;
    }


    // ---- tasks (off-diagonal pair, side): two rounds of wavefronts per pair pair, then the combination ----------------
    for (int pq0 = 0; pq0 < Poff; pq0 += NW / 2) {
        // D = 3: three pairs for two pairs of wavefronts -- in the second round BOTH pairs of wavefronts take the last pair, each one
        // half of its point chunks (the moment matrices are sums over points: the halves are added before the combination),
        // instead of one pair of wavefronts idling through it
        const bool split = sep_grad_version(DP) == 1 && NW == 4 && Poff - pq0 == 1;
        const int half = split ? (wave >> 1) : 0;
        const int pq = split ? pq0 : pq0 + (wave >> 1), side = wave & 1;
        const bool active = pq < Poff && s_K[pq < Poff ? pq : 0] > 0;
        const int K = active ? s_K[pq] : 0;
        const int C = p.mono_cum[K];
        const int nb = (C + 15) >> 4;
        if (active) {
            const int qf = s_q[pq];
            int a = 0, rem = qf;
            while (rem >= D - a) { rem -= D - a; ++a; }
            const int b = a + rem;
            const int co = side ? b : a;
            const double* Z = s_Z + pq * DP * DP;
            const double* il = s_ils2 + co * E;
            const double lv = p.logvar[co];
            mfma_d4 acc[NA][NB];
            double accE[NE > 0 ? NE : 1][NB];
            if constexpr (sep_grad_version(DP) == 1) {
            const int K1 = K + 1;
            // word offsets of the lane's monomial factors inside a point's table (loop-invariant: one address add per LDS read in
            // the k-steps; forming them from the packed exponents there cost 3 - 4 integer instructions per read, 122 VALU
            // instructions per 4-point step by the counters against ~16 matrix instructions)
            const int zslot = 1 + NV + D * K1;                 // the zero of a point's table
            int offB[NB][DP];
#pragma unroll
            for (int ib = 0; ib < NB; ++ib) {
                const int ex = s_me[ib * 16 + r16];
#pragma unroll
                for (int d = 0; d < DP; ++d) offB[ib][d] = 1 + NV + d * K1 + ((ex >> (8 * d)) & 255);
                if (ib * 16 + r16 >= C) offB[ib][0] = zslot;      // monomial slots past the degree's count contribute nothing
            }
            // weightings: wt * v[s1] * v[s2]; rows past the last weighting read the zero
            int offA[NA][2], offE[NE > 0 ? NE : 1][2];
#pragma unroll
            for (int ia = 0; ia < NA; ++ia) { offA[ia][0] = selA[ia][0] >= 0 ? 1 + selA[ia][0] : zslot; offA[ia][1] = 1 + selA[ia][1]; }
#pragma unroll
            for (int e = 0; e < NE; ++e) { offE[e][0] = selE[e][0] >= 0 ? 1 + selE[e][0] : zslot; offE[e][1] = 1 + selE[e][1]; }
#pragma unroll
            for (int ia = 0; ia < NA; ++ia)
#pragma unroll
                for (int ib = 0; ib < NB; ++ib) acc[ia][ib] = mfma_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int e = 0; e < NE; ++e)
#pragma unroll
                for (int ib = 0; ib < NB; ++ib) accE[e][ib] = 0.0;
            const int nch = (N + 63) >> 6, nch0 = (nch + 1) >> 1;
            const int c_lo = split ? half * nch0 * 64 : 0;
            const int c_hi = (split && half == 0) ? (nch0 * 64 < N ? nch0 * 64 : N) : N;
            for (int c0 = c_lo; c0 < c_hi; c0 += 64) {
                // -- per-point tables (lane = point) --
                {
                    const int pt0 = c0 + lane;
                    const bool live = pt0 < N;
                    const int pt = live ? pt0 : N - 1;
                    double* tp = tabw + (size_t)lane * PS;
                    double nu[DP], x[DP], zx[DP];
                    double ks = 0.0;
#pragma unroll
                    for (int d = 0; d < DP; ++d) {
                        nu[d] = (d < D) ? p.Xt[(size_t)d * N + pt] - s_m[d] : 0.0;
                        x[d] = (d < D) ? nu[d] * il[d] : 0.0;          // u (rows) or w (columns)
                        ks = fma(nu[d], x[d], ks);
                    }
                    tp[1] = 1.0;
                    tp[zslot] = 0.0;
#pragma unroll
                    for (int d = 0; d < DP; ++d) if (d < D) tp[2 + d] = x[d];
                    for (int xx = 0; xx < NX; ++xx) {
                        const double v = p.Xt[(size_t)(D + xx) * N + pt] - s_m[D + xx];
                        ks = fma(v * v, il[D + xx], ks);
                        tp[2 + D + xx] = v;
                    }
                    double qq = 0.0;
                    double g[DP];
#pragma unroll
                    for (int d = 0; d < DP; ++d) g[d] = 0.0;
#pragma unroll
                    for (int i = 0; i < DP; ++i) {
                        zx[i] = 0.0;
#pragma unroll
                        for (int j = 0; j < DP; ++j) {
                            zx[i] = fma(Z[i * DP + j], x[j], zx[i]);
                            g[j] = fma(Z[i * DP + j], x[i], g[j]);            // Z^T x
                        }
                        qq = fma(x[i], zx[i], qq);
                    }
                    const double kk = lv - 0.5 * ks + 0.5 * qq;
                    tp[0] = live ? fast_exp(kk, s_tab) * p.beta[(size_t)co * N + pt] : 0.0;
                    // monomial variables: g = Z^T u on the row side, w itself on the column side
                    double* pw = tp + 1 + NV;
#pragma unroll
                    for (int d = 0; d < DP; ++d) {
                        if (d < D) {
                            const double xv = side ? x[d] : g[d];
                            double pv = 1.0;
                            for (int e = 0; e < K1; ++e) { pw[d * K1 + e] = pv; pv *= xv; }
                        }
                    }
                }
                wave_lds_sync();
                // -- k-steps of 4 points (16 for a full chunk; the ragged last chunk only as many as it has points: at N = 200 its 8
                //    points took 16 steps, 64 instead of 50 per task): F^T Phi on the matrix cores.  Branch-free for a compile-time
                //    block count: every LDS read of a step can be in flight before the first product (the first version branched per
                //    block and per factor, each read followed by its wait: 15 % matrix-pipe and 32 % vector utilisation by the counters) --
                const int nks = ((N - c0 < 64 ? N - c0 : 64) + 3) >> 2;
                auto ksteps = [&](auto nbc) {
                    constexpr int NBK = decltype(nbc)::value;
#pragma unroll 1
                    for (int ks4 = 0; ks4 < nks; ++ks4) {
                        const double* tp = tabw + (size_t)(ks4 * 4 + kq) * PS;
                        const double wt = tp[0];
                        double aF[NA], aE[NE > 0 ? NE : 1], phi[NBK];
#pragma unroll
                        for (int ia = 0; ia < NA; ++ia) aF[ia] = wt * tp[offA[ia][0]] * tp[offA[ia][1]];
#pragma unroll
                        for (int e = 0; e < NE; ++e) aE[e] = wt * tp[offE[e][0]] * tp[offE[e][1]];
#pragma unroll
                        for (int ib = 0; ib < NBK; ++ib) {
                            phi[ib] = tp[offB[ib][0]];
#pragma unroll
                            for (int d = 1; d < DP; ++d) phi[ib] *= tp[offB[ib][d]];
                        }
#pragma unroll
                        for (int ib = 0; ib < NBK; ++ib) {
#pragma unroll
                            for (int ia = 0; ia < NA; ++ia) acc[ia][ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(aF[ia], phi[ib], acc[ia][ib], 0, 0, 0);
#pragma unroll
                            for (int e = 0; e < NE; ++e) accE[e][ib] = fma(aE[e], phi[ib], accE[e][ib]);
                        }
                    }
                };
                static_assert(NB == 5, "block-count dispatch below");
                switch (nb) {
                    case 1: ksteps(std::integral_constant<int, 1>{}); break;
                    case 2: ksteps(std::integral_constant<int, 2>{}); break;
                    case 3: ksteps(std::integral_constant<int, 3>{}); break;
                    case 4: ksteps(std::integral_constant<int, 4>{}); break;
                    default: ksteps(std::integral_constant<int, 5>{}); break;
                }
                wave_lds_sync();
            }
            } else {
            const int K1 = K + 1;
            constexpr int FW = 16 * NA + NE;                   // weightings kept per point
            constexpr int NV0 = DP >= 3 ? 2 : 1, NV1 = DP - NV0;          // monomial variables of the two factor tables (D == DP here)
            const int n0 = sep_grad_tri(K1, NV0);
            // word offsets of the lane's two monomial factors inside a point's table (loop-invariant)
            auto tri_off = [&](int ea, int eb) { return ea * K1 - (ea * (ea - 1)) / 2 + eb; };      // ea + eb <= K
            int offB[NB][2];
#pragma unroll
            for (int ib = 0; ib < NB; ++ib) {
                const int ex = s_me[ib * 16 + r16];
                const int e0 = ex & 255, e1 = (ex >> 8) & 255, e2 = (ex >> 16) & 255, e3 = (ex >> 24) & 255;
                int o0, o1;
                if (DP == 2) { o0 = e0; o1 = e1; }
                else if (DP == 3) { o0 = tri_off(e0, e1); o1 = e2; }
                else { o0 = tri_off(e0, e1); o1 = tri_off(e2, e3); }
                // monomial slots past the degree's count are never read by the combination (it stops at C): any finite value will do
                // -- the weighting wt itself (their exponents may exceed K: the factor tables have no such entry)
                const bool in = ib * 16 + r16 < C;
                offB[ib][0] = in ? FW + o0 : 0;
                offB[ib][1] = in ? FW + n0 + o1 : 0;
            }
#pragma unroll
            for (int ia = 0; ia < NA; ++ia)
#pragma unroll
                for (int ib = 0; ib < NB; ++ib) acc[ia][ib] = mfma_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int e = 0; e < NE; ++e)
#pragma unroll
                for (int ib = 0; ib < NB; ++ib) accE[e][ib] = 0.0;
            // table pass roles: two lanes per point -- both form the weightings (the lower one stores them), each one of the two
            // factor tables (uniform code, the variables chosen by data)
            constexpr int kSepGradChunk = sep_grad_chunk(DP);
            const int tpt = lane & (kSepGradChunk - 1), half = kSepGradChunk == 64 ? 0 : lane >> 5;
            // The inputs of a chunk's points are loaded ONE CHUNK AHEAD, unconditionally and from clamped indices: a load behind a
            // per-lane condition (`d < D ? X[..] : 0`) compiles to an exec-masked branch with its own wait, i.e. the L2 latencies of
            // the D + NX + 1 loads in sequence at the head of every chunk -- with 32-point chunks that was most of the pass (first
            // build of this version: 87 ms at config 4 against 66 for its predecessor, profiles/r04e_sepv2a_*).
            constexpr int NXL = 6;                             // extra inputs (actions + time) the gradient kernels take
            // the task's small operands (Z, 1 / l^2, the input mean) as wave-uniform values in scalar registers: as LDS reads they sit
            // behind the wavefront fences of the chunk loop and were re-read -- 16 + 2 (D + NX) broadcast reads and their latency --
            // at the head of every chunk
            auto uni = [](double v) {
                return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
            };
            double Zr[DP][DP], ilr[DP + NXL], smr[DP + NXL];
#pragma unroll
            for (int i = 0; i < DP; ++i)
#pragma unroll
                for (int j = 0; j < DP; ++j) Zr[i][j] = uni(Z[i * DP + j]);
#pragma unroll
            for (int e = 0; e < DP + NXL; ++e) {
                const int ee = e < DP ? (e < D ? e : 0) : D + (e - DP < NX ? e - DP : 0);
                const bool in = e < DP ? e < D : e - DP < NX;
                ilr[e] = in ? uni(il[ee]) : 0.0;
                smr[e] = in ? uni(s_m[ee]) : 0.0;
            }
            double xin[DP], xein[NXL], bin;
            auto load_chunk = [&](int c0) {
                const int pt0 = c0 + tpt;
                const int pt = pt0 < N ? pt0 : N - 1;
#pragma unroll
                for (int d = 0; d < DP; ++d) xin[d] = p.Xt[(size_t)(d < D ? d : D - 1) * N + pt];
#pragma unroll
                for (int xx = 0; xx < NXL; ++xx) xein[xx] = p.Xt[(size_t)(D + (xx < NX ? xx : 0)) * N + pt];     // NX >= 1 (an action)
                bin = p.beta[(size_t)co * N + pt];
            };
            load_chunk(0);
            for (int c0 = 0; c0 < N; c0 += kSepGradChunk) {
                // -- per-point tables --
                {
                    const int pt0 = c0 + tpt;
                    const bool live = pt0 < N;
                    double* tp = tabw + (size_t)tpt * PS;
                    double nu[DP], x[DP], zx[DP], xe[NXL];
                    double ks = 0.0;
#pragma unroll
                    for (int d = 0; d < DP; ++d) {
                        nu[d] = (d < D) ? xin[d] - smr[d] : 0.0;
                        x[d] = nu[d] * ilr[d];                         // u (rows) or w (columns)
                        ks = fma(nu[d], x[d], ks);
                    }
#pragma unroll
                    for (int xx = 0; xx < NXL; ++xx) {
                        xe[xx] = (xx < NX) ? xein[xx] - smr[DP + xx] : 0.0;
                        ks = fma(xe[xx] * xe[xx], ilr[DP + xx], ks);
                    }
                    const double bpt = bin;
                    if (c0 + kSepGradChunk < N) load_chunk(c0 + kSepGradChunk);          // in flight during this chunk's tables and k-steps
                    double qq = 0.0;
                    double g[DP];
#pragma unroll
                    for (int d = 0; d < DP; ++d) g[d] = 0.0;
#pragma unroll
                    for (int i = 0; i < DP; ++i) {
                        zx[i] = 0.0;
#pragma unroll
                        for (int j = 0; j < DP; ++j) {
                            zx[i] = fma(Zr[i][j], x[j], zx[i]);
                            g[j] = fma(Zr[i][j], x[i], g[j]);                  // Z^T x
                        }
                        qq = fma(x[i], zx[i], qq);
                    }
                    const double kk = lv - 0.5 * ks + 0.5 * qq;
                    const double wt = live ? fast_exp(kk, s_tab) * bpt : 0.0;
                    // weightings wt f(v), f in {1, v_d, v_d v_e (d <= e), nu_x}: compile-time positions (D == DP)
                    if (half == 0) {
                        constexpr int T2c = DP * (DP + 1) / 2;
                        tp[0] = wt;
                        double wv[DP];
#pragma unroll
                        for (int d = 0; d < DP; ++d) { wv[d] = wt * x[d]; if (1 + d < FW) tp[1 + d] = wv[d]; }
                        int r = 1 + DP;
#pragma unroll
                        for (int d = 0; d < DP; ++d)
#pragma unroll
                            for (int e = d; e < DP; ++e) { if (r < FW) tp[r] = wv[d] * x[e]; ++r; }
#pragma unroll
                        for (int xx = 0; xx < NXL; ++xx) { if (1 + DP + T2c + xx < FW) tp[1 + DP + T2c + xx] = wt * xe[xx]; }     // zero past NX
#pragma unroll
                        for (int r2 = 1 + DP + T2c + NXL; r2 < FW; ++r2) tp[r2] = 0.0;
                    }
                    // factor tables x_a^i x_b^j, i + j <= K, of the two variable groups (monomial variables: g = Z^T u on the row side, w
                    // itself on the column side): both by this lane (64-point chunks) or the group of its half (32-point chunks)
                    {
                        double mv[DP];
#pragma unroll
                        for (int d = 0; d < DP; ++d) mv[d] = side ? x[d] : g[d];
#pragma unroll
                        for (int hh = 0; hh < (kSepGradChunk == 64 ? 2 : 1); ++hh) {
                            const int hs = kSepGradChunk == 64 ? hh : half;
                            double xa, xb;
                            if (DP == 2) { xa = hs ? mv[1] : mv[0]; xb = 0.0; }
                            else if (DP == 3) { xa = hs ? mv[2] : mv[0]; xb = hs ? 0.0 : mv[1]; }
                            else { xa = hs ? mv[DP >= 4 ? 2 : 0] : mv[0]; xb = hs ? mv[DP >= 4 ? 3 : 0] : mv[1]; }
                            const bool two = hs ? (NV1 == 2) : (NV0 == 2);
                            double* tq = tp + FW + (hs ? n0 : 0);
                            double pa = 1.0;
                            int o = 0;
                            for (int ea = 0; ea < K1; ++ea) {
                                double pab = pa;
                                const int nbj = two ? K1 - ea : 1;
                                for (int eb = 0; eb < nbj; ++eb) { tq[o++] = pab; pab *= xb; }
                                pa *= xa;
                            }
                        }
                    }
                }
                wave_lds_sync();
                // -- k-steps of 4 points: F^T Phi on the matrix cores; branch-free for a compile-time block count --
                auto ksteps = [&](auto nbc) {
                    constexpr int NBK = decltype(nbc)::value;
                    // the operands of step s + 1 are read before the matrix instructions of step s issue (a wavefront issues in order:
                    // behind three 64-cycle matrix instructions the reads of the next step would start 130 cycles of LDS latency late)
                    double aF[NA], aE[NE > 0 ? NE : 1], b0[NBK], b1[NBK];
                    auto fetch = [&](int ks4, double (&fa)[NA], double (&fe)[NE > 0 ? NE : 1], double (&f0)[NBK], double (&f1)[NBK]) {
                        const double* tp = tabw + (size_t)(ks4 * 4 + kq) * PS;
#pragma unroll
                        for (int ia = 0; ia < NA; ++ia) fa[ia] = tp[16 * ia + r16];
#pragma unroll
                        for (int e = 0; e < NE; ++e) fe[e] = tp[16 * NA + e];
#pragma unroll
                        for (int ib = 0; ib < NBK; ++ib) { f0[ib] = tp[offB[ib][0]]; f1[ib] = tp[offB[ib][1]]; }
                    };
                    fetch(0, aF, aE, b0, b1);
#pragma unroll 1
                    for (int ks4 = 0; ks4 < kSepGradChunk / 4; ++ks4) {
                        double nF[NA], nE[NE > 0 ? NE : 1], n0[NBK], n1[NBK];
                        fetch(ks4 + 1 < kSepGradChunk / 4 ? ks4 + 1 : ks4, nF, nE, n0, n1);
                        double phi[NBK];
#pragma unroll
                        for (int ib = 0; ib < NBK; ++ib) phi[ib] = b0[ib] * b1[ib];
#pragma unroll
                        for (int ib = 0; ib < NBK; ++ib) {
#pragma unroll
                            for (int ia = 0; ia < NA; ++ia) acc[ia][ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(aF[ia], phi[ib], acc[ia][ib], 0, 0, 0);
#pragma unroll
                            for (int e = 0; e < NE; ++e) accE[e][ib] = fma(aE[e], phi[ib], accE[e][ib]);
                        }
#pragma unroll
                        for (int ia = 0; ia < NA; ++ia) aF[ia] = nF[ia];
#pragma unroll
                        for (int e = 0; e < NE; ++e) aE[e] = nE[e];
#pragma unroll
                        for (int ib = 0; ib < NBK; ++ib) { b0[ib] = n0[ib]; b1[ib] = n1[ib]; }
                    }
                };
                static_assert(NB == 5, "block-count dispatch below");
                switch (nb) {
                    case 1: ksteps(std::integral_constant<int, 1>{}); break;
                    case 2: ksteps(std::integral_constant<int, 2>{}); break;
                    case 3: ksteps(std::integral_constant<int, 3>{}); break;
                    case 4: ksteps(std::integral_constant<int, 4>{}); break;
                    default: ksteps(std::integral_constant<int, 5>{}); break;
                }
                wave_lds_sync();
            }
            }
            // moment matrix of this (pair, side) into the wavefront's region: M[weighting][monomial], row stride 16 NB
#pragma unroll
            for (int ia = 0; ia < NA; ++ia)
#pragma unroll
                for (int ib = 0; ib < NB; ++ib)
                    if (ib < nb) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (ia * 16 + 4 * r + kq < nW) tabw[(size_t)(ia * 16 + 4 * r + kq) * (16 * NB) + ib * 16 + r16] = acc[ia][ib][r];
                    }
            // extra weightings: a lane holds the sum over ITS quarter of the points (kq) -> the four quarters, fixed order
#pragma unroll
            for (int e = 0; e < NE; ++e)
#pragma unroll
                for (int ib = 0; ib < NB; ++ib)
                    if (ib < nb && NA * 16 + e < nW) {
                        const double h = swap_add<16>(accE[e][ib], accE[e][ib]);
                        const double tot = swap_add<32>(h, h);
                        if (kq == 0) tabw[(size_t)(NA * 16 + e) * (16 * NB) + ib * 16 + r16] = tot;
                    }
        }
        __syncthreads();
        if (split && active) {                       // (uniform over the workgroup: all four wavefronts worked on pair pq0)
            const int nmat = nW * 16 * NB;
            for (int idx = tid; idx < 2 * nmat; idx += NT) {
                const int sd = idx >= nmat ? 1 : 0, k = idx - sd * nmat;
                double* dst = s_wave + (size_t)sd * p.wave_words;
                dst[k] += dst[(size_t)2 * p.wave_words + k];
            }
            __syncthreads();
        }
        // ---- combination: the two wavefronts of a pair share the outputs (side 0 wave: even outputs, side 1: odd) --------
        if (active && half == 0) {
            const int qf = s_q[pq];
            int a = 0, rem = qf;
            while (rem >= D - a) { rem -= D - a; ++a; }
            const int b = a + rem;
            const double* G = s_wave + (size_t)(wave & ~1) * p.wave_words;           // side 0: rows (u)
            const double* Hm = G + p.wave_words;                                     // side 1: columns (w)
            const int RSm = 16 * NB;
            double* out = p.mom + (((size_t)c * H + t) * P + qf) * p.NSP;
            // weighting indices: 0: 1 | 1 + d: u_d | 1 + D + tri(d, e): u_d u_e | 1 + D + D (D + 1) / 2 + x: nu_x
            auto widx_uu = [&](int d, int e) { return 1 + D + d * D - (d * (d - 1)) / 2 + (e - d); };
            for (int o = side; o < p.NSP; o += 2) {
                // which moment: [W | P1 (DP) | P2 tri (NH, DP-padded index) | Pe (NXP)]
                int kind, d = 0, e = 0;
                if (o == 0) kind = 0;
                else if (o < 1 + DP) { kind = 1; d = o - 1; }
                else if (o < 1 + DP + NH) { kind = 2; decode_tri(o - 1 - DP, DP, d, e); }
                else { kind = 3; d = o - 1 - DP - NH; }
                double v = 0.0;
                const bool valid = (kind == 0) || (kind == 1 && d < D) || (kind == 2 && d < D && e < D) || (kind == 3 && d < NX);
                if (valid) {
                    for (int n = lane; n < C; n += 64) {
                        const double cw = s_cw[n];
                        double s;
                        if (kind == 0) s = G[n] * Hm[n];
                        else if (kind == 1) s = G[(1 + d) * RSm + n] * Hm[n] + G[n] * Hm[(1 + d) * RSm + n];
                        else if (kind == 2) s = G[widx_uu(d, e) * RSm + n] * Hm[n] + G[(1 + d) * RSm + n] * Hm[(1 + e) * RSm + n]
                                                + G[(1 + e) * RSm + n] * Hm[(1 + d) * RSm + n] + G[n] * Hm[widx_uu(d, e) * RSm + n];
                        else {
                            const int wx = 1 + D + D * (D + 1) / 2 + d;
                            s = G[wx * RSm + n] * Hm[n] * s_ils2[a * E + D + d] + G[n] * Hm[wx * RSm + n] * s_ils2[b * E + D + d];
                        }
                        v = fma(cw, s, v);
                    }
                    v = wave_sum(v);
                }
                if (lane == 0) out[o] = v;
            }
        }
        __syncthreads();
    }
}

}  // namespace gpmpc_hip
