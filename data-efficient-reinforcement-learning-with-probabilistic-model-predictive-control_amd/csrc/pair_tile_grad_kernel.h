// pair_tile_grad_kernel.h -- moments of the DIAGONAL output pairs for the analytic gradient, batch-major (gfx950 / MI355X).
//
// After grad_sep_kernel.h took the off-diagonal pairs, what the element-wise moment pass still does at config 4 is the
// four diagonal pairs -- and it re-streams the tables T_a = beta_a beta_a^T - iK_a through the fabric once per
// (candidate, horizon step) exactly as the forward rollout did before pair_tile_kernel.h (495 ms of a 706 ms gradient).
// With the trajectory known all (candidate, step) items are independent, so the same batch-major structure applies to the
// whole batch at once: a workgroup owns a 128 x 128 tile of T_a in registers and loops over a chunk of ITEMS (c, t).
//
// A diagonal pair is symmetric, E_ij = E_ji with E_ij = T_ij ea_i ea_j P_K(g_i . u_j), and only i <= j is stored (diagonal
// halved), so with  c_j = sum_{i<=j} E~_ij,  r_i = sum_{j>=i} E~_ij,  V_j = sum_{i<=j} E~_ij u_i  the moments of
// p_ij = u_i + u_j over the FULL square are
//     W  = 2 sum_j c_j                       P1 = 2 sum_j (V_j + c_j u_j)
//     P2 = 2 [ sum_j (c_j u_j u_j^T + V_j u_j^T + u_j V_j^T) + sum_i r_i u_i u_i^T ]
//     Pe = 2 / l_ax^2 [ sum_j c_j nu_jx + sum_i r_i nu_ix ]
// i.e. per element, on top of the forward's E: one add into the column sum, D FMAs into V_j (lanes own columns: registers),
// and one FMA into the row sum of the lane's row slot (16 per lane, reduced over the lanes once per item).  Items whose
// Taylor degree is 0 (direct-exp form) are left to the element-wise kernel.  Per-tile partial moments go to HBM;
// tile_moments_reduce_kernel adds the tiles in a fixed order, doubles, and writes the pair's slot of the moment array.
#pragma once
#include "pair_tile_kernel.h"
#include "grad_kernels.h"

namespace gpmpc_hip {

constexpr int kTgMom = 24;            // partial moments per (item, a, tile): W | P1 (DP) | P2 upper triangle | Pe (<= 8 action / time inputs)

struct TileGradLayout {
    int tab, xs, rows, cols, wsum, rsum, total;
};

__host__ __device__ inline TileGradLayout make_tile_grad_layout(int DP, int E) {
    TileGradLayout L;
    int o = 0;
    L.tab = o;  o += 64;
    L.xs = o;   o += 2 * E * kTileW;                                   // [side][e][point]
    L.rows = o; o += 2 * kTileGC * kTileW * (2 * DP + 2);              // [buffer][item of the group][row][ev, g (DP), u (DP), pad]
    L.cols = o; o += 2 * kTileGC * (DP + 1) * kTileW;                  // [buffer][item][component][column]
    L.wsum = o; o += 2 * kTileGC * kTileWaves * kTgMom;                // [buffer][item][wavefront][moment]
    L.rsum = o; o += kTileWaves * kTileRW;                             // per wavefront: the row sums of its 16 rows
    L.total = o;
    return L;
}

// 16 rows x 2 columns per lane with the moment accumulators.  rec: {ev, g (DP), u (DP)} per row, stride RSG.
// Two rows per iteration (four independent Taylor chains) and the records of the NEXT two rows fetched from the LDS before the
// current ones are used: the kernel runs at 2 wavefronts per SIMD (~200 VGPRs), so the latency of an LDS read or of a
// dependent fp64 FMA is not covered by other wavefronts (first version, read-then-wait per row: 126 ms per 30720 items of
// config 4 against 65 ms for the forward's tile kernel on the same elements).
template <int DP, int K, int RSG>
__device__ inline void tile_rows_moments(const double* rec, const double (&tv)[kTileRW][2], const double (&u0)[DP], const double (&u1)[DP],
                                         double f0, double f1, double& cs0, double& cs1, double (&v0)[DP], double (&v1)[DP],
                                         double (&racc)[kTileRW]) {
    constexpr int RL = 1 + 2 * DP;
    double qa[RL], qb[RL], na[RL], nb[RL];
#pragma unroll
    for (int k = 0; k < RL; ++k) { qa[k] = rec[k]; qb[k] = rec[RSG + k]; }
#pragma unroll
    for (int r = 0; r < kTileRW; r += 2) {
        if (r + 2 < kTileRW) {
#pragma unroll
            for (int k = 0; k < RL; ++k) { na[k] = rec[(r + 2) * RSG + k]; nb[k] = rec[(r + 3) * RSG + k]; }
        }
        double c00 = qa[1] * u0[0], c01 = qa[1] * u1[0], c10 = qb[1] * u0[0], c11 = qb[1] * u1[0];
#pragma unroll
        for (int d = 1; d < DP; ++d) {
            c00 = fma(qa[1 + d], u0[d], c00);
            c01 = fma(qa[1 + d], u1[d], c01);
            c10 = fma(qb[1 + d], u0[d], c10);
            c11 = fma(qb[1 + d], u1[d], c11);
        }
        const double ta0 = qa[0] * tv[r][0], ta1 = qa[0] * tv[r][1], tb0 = qb[0] * tv[r + 1][0], tb1 = qb[0] * tv[r + 1][1];
        const double e00 = taylor_exp<K>(c00) * ta0;                  // E~_ij without the column factor ea_j
        const double e01 = taylor_exp<K>(c01) * ta1;
        const double e10 = taylor_exp<K>(c10) * tb0;
        const double e11 = taylor_exp<K>(c11) * tb1;
        cs0 += e00; cs1 += e01;
        cs0 += e10; cs1 += e11;
#pragma unroll
        for (int d = 0; d < DP; ++d) {
            v0[d] = fma(e00, qa[1 + DP + d], v0[d]);
            v1[d] = fma(e01, qa[1 + DP + d], v1[d]);
            v0[d] = fma(e10, qb[1 + DP + d], v0[d]);
            v1[d] = fma(e11, qb[1 + DP + d], v1[d]);
        }
        racc[r] = fma(e00, f0, fma(e01, f1, racc[r]));
        racc[r + 1] = fma(e10, f0, fma(e11, f1, racc[r + 1]));
        if (r + 2 < kTileRW) {
#pragma unroll
            for (int k = 0; k < RL; ++k) { qa[k] = na[k]; qb[k] = nb[k]; }
        }
    }
}

// FUSED: launched by the batch-major FORWARD of a gradient call, once per horizon step (items = the candidates, records = the
// step's own records): the tile sums the forward needs (p.part, as pair_tile_kernel writes them) are the W moment, so the
// forward's tile pass is not repeated by a separate moment pass over the stored trajectory -- the same E_ij would otherwise be
// evaluated twice.  Direct-exp items (degree 0) get their forward sum here too (tile_rows_exp) and stay with the element-wise
// moment kernel for their moments.
template <int DP, bool FUSED = false>
__global__ __launch_bounds__(kTileWaves * 64, 2) void pair_tile_moments_kernel(const StepArgs p, double* __restrict__ tmom) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int RSG = 2 * DP + 2;
    constexpr int NTH = kTileWaves * 64;
    constexpr int NH = DP * (DP + 1) / 2;
    constexpr int NXC = (kTgMom - 1 - DP - NH) < 8 ? (kTgMom - 1 - DP - NH) : 8;      // action / time inputs the partial-moment block has room for
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = p.N, D = p.D, E = p.E, NX = E - D;

    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nta = p.ntiles * D;
    const int cnt = (nta - xcd + 7) >> 3;
    if (cnt <= 0 || slot >= cnt * p.nchunk) return;
    const int ta = xcd + 8 * (slot / p.nchunk);
    const int chunk = slot - (slot / p.nchunk) * p.nchunk;
    const int a = ta / p.ntiles;
    const int tile = ta - a * p.ntiles;
    int rb = 0, kcol = tile;
    while (kcol >= p.nb - rb) { kcol -= p.nb - rb; ++rb; }
    const int cb = rb + kcol;
    const int i0 = rb * kTileW, j0 = cb * kTileW;
    const int c0 = chunk * p.cch;
    const int c1 = (c0 + p.cch < p.B) ? c0 + p.cch : p.B;
    const int ncand = c1 - c0;

    const TileGradLayout L = make_tile_grad_layout(DP, E);
    double* s_tab = smem + L.tab;
    double* s_xs = smem + L.xs;
    double* s_rows = smem + L.rows;
    double* s_cols = smem + L.cols;
    double* s_wsum = smem + L.wsum;
    double* s_rsum = smem + L.rsum + wave * kTileRW;

    double tv[kTileRW][2];
    {
        const double* Ta = p.Tm + (size_t)a * (N + kTPad) * N;
        const int j = j0 + 2 * lane;
#pragma unroll
        for (int r = 0; r < kTileRW; ++r) {
            const int i = i0 + wave * kTileRW + r;
            const bool ri = i < N;
            tv[r][0] = (ri && j < N) ? Ta[(size_t)i * N + j] : 0.0;
            tv[r][1] = (ri && j + 1 < N) ? Ta[(size_t)i * N + j + 1] : 0.0;
        }
    }
    for (int k = tid; k < 64; k += NTH) s_tab[k] = kExp2Tab[k];
    for (int k = tid; k < 2 * E * kTileW; k += NTH) {
        const int side = k / (E * kTileW);
        const int rem = k - side * (E * kTileW);
        const int e = rem / kTileW, pt = rem - e * kTileW;
        int gp = (side ? j0 : i0) + pt;
        gp = gp < N ? gp : N - 1;
        s_xs[k] = p.Xt[(size_t)e * N + gp];
    }
    __syncthreads();

    const double* il = p.ils2 + (size_t)a * E;
    const double lv = p.logvar[a];
    const int ngroups = (ncand + kTileGC - 1) / kTileGC;
    // compact records hold the D diagonal pairs only; the forward's step records all P pairs
    const double* __restrict__ tpar = p.crec + p.off_pair + (p.compact ? a : pair_index(a, a, p.D)) * p.PRP;
    auto degrees = [&](int g, int (&K)[kTileGC]) {
#pragma unroll
        for (int kk = 0; kk < kTileGC; ++kk) {
            const int cl = g * kTileGC + kk;
            K[kk] = (cl < ncand) ? ((int)tpar[(size_t)(c0 + cl) * p.CS + DP * DP + 1] & 63) : 0;
        }
    };

    auto records = [&](int g, const int (&Kg)[kTileGC]) {
        const int kk = wave >> 2;
        const int cl = g * kTileGC + kk;
        const int K = kk ? Kg[1] : Kg[0];
        if (cl >= ncand || (K == 0 && !FUSED)) return;
        const double* par = tpar + (size_t)(c0 + cl) * p.CS;
        const double* mo = p.crec + (size_t)(c0 + cl) * p.CS;
        const int side = (wave >> 1) & 1;
        const int pt = (wave & 1) * 64 + lane;
        const double* xp = s_xs + (size_t)side * E * kTileW + pt;
        const double* Qs = par + DP * DP + 2;
        const double* Gs = Qs + DP * DP;
        double nu[DP], u[DP], g_[DP];
        double qf = 0.0;
#pragma unroll
        for (int d = 0; d < DP; ++d) {
            nu[d] = (d < D) ? xp[d * kTileW] - mo[d] : 0.0;
            u[d] = nu[d] * ((d < D) ? il[d] : 0.0);
        }
        for (int e = D; e < E; ++e) {
            const double v = xp[e * kTileW] - mo[e];
            qf = fma(v * v, il[e], qf);
        }
#pragma unroll
        for (int i = 0; i < DP; ++i) {
            double r = 0.0, gi = 0.0;
#pragma unroll
            for (int j = 0; j < DP; ++j) {
                r = fma(Qs[i * DP + j], nu[j], r);
                gi = fma(Gs[i * DP + j], nu[j], gi);
            }
            qf = fma(nu[i], r, qf);
            g_[i] = gi;
        }
        const double kkv = fma(-0.5, qf, lv);
        const double f = (K > 0) ? fast_exp(kkv, s_tab) : kkv;       // degree 0 (FUSED only): the log-factor, as pair_tile_kernel
        const int buf = g & 1;
        if (side == 0) {
            double* rec = s_rows + ((size_t)(buf * kTileGC + kk) * kTileW + pt) * RSG;
            rec[0] = f;
#pragma unroll
            for (int d = 0; d < DP; ++d) { rec[1 + d] = g_[d]; rec[1 + DP + d] = u[d]; }
        } else {
            double* col = s_cols + (size_t)(buf * kTileGC + kk) * (DP + 1) * kTileW + pt;
            col[0] = f;
#pragma unroll
            for (int d = 0; d < DP; ++d) col[(1 + d) * kTileW] = u[d];
        }
    };

    auto pairs = [&](int g, const int (&Kg)[kTileGC]) {
        const int buf = g & 1;
#pragma unroll 1
        for (int kk = 0; kk < kTileGC; ++kk) {
            const int cl = g * kTileGC + kk;
            if (cl >= ncand) break;
            const int K = kk ? Kg[1] : Kg[0];
            double* wout = s_wsum + (size_t)((buf * kTileGC + kk) * kTileWaves + wave) * kTgMom;
            if (K == 0 && !FUSED) continue;                         // direct-exp item: the element-wise kernel's
            const double* mo = p.crec + (size_t)(c0 + cl) * p.CS;
            const double* col = s_cols + (size_t)(buf * kTileGC + kk) * (DP + 1) * kTileW + 2 * lane;
            const double f0 = col[0], f1 = col[1];
            double u0[DP], u1[DP];
#pragma unroll
            for (int d = 0; d < DP; ++d) { u0[d] = col[(1 + d) * kTileW]; u1[d] = col[(1 + d) * kTileW + 1]; }
            const double* rec = s_rows + ((size_t)(buf * kTileGC + kk) * kTileW + wave * kTileRW) * RSG;
            if constexpr (FUSED) {
                if (K == 0) {
                    // forward sum only (slot 0 of the wavefront's partial moments = what pair_tile_kernel writes); moments: element-wise
                    double s0, s1;
                    tile_rows_exp<DP, RSG>(rec, tv, u0, u1, f0, f1, s_tab, s0, s1);
                    const double v = wave_sum(s0 + s1);
                    if (lane == 0) wout[0] = v;
                    continue;
                }
            }
            double cs0 = 0.0, cs1 = 0.0, v0[DP], v1[DP], racc[kTileRW];
#pragma unroll
            for (int d = 0; d < DP; ++d) { v0[d] = 0.0; v1[d] = 0.0; }
#pragma unroll
            for (int r = 0; r < kTileRW; ++r) racc[r] = 0.0;
            if (K <= 2) tile_rows_moments<DP, 2, RSG>(rec, tv, u0, u1, f0, f1, cs0, cs1, v0, v1, racc);
            else if (K == 3) tile_rows_moments<DP, 3, RSG>(rec, tv, u0, u1, f0, f1, cs0, cs1, v0, v1, racc);
            else if (K == 4) tile_rows_moments<DP, 4, RSG>(rec, tv, u0, u1, f0, f1, cs0, cs1, v0, v1, racc);
            else if (K <= 6) tile_rows_moments<DP, 6, RSG>(rec, tv, u0, u1, f0, f1, cs0, cs1, v0, v1, racc);
            else if (K <= 8) tile_rows_moments<DP, 8, RSG>(rec, tv, u0, u1, f0, f1, cs0, cs1, v0, v1, racc);
            else if (K <= 10) tile_rows_moments<DP, 10, RSG>(rec, tv, u0, u1, f0, f1, cs0, cs1, v0, v1, racc);
            else tile_rows_moments<DP, 14, RSG>(rec, tv, u0, u1, f0, f1, cs0, cs1, v0, v1, racc);
            // column factor ea_j on the column-side sums
            cs0 *= f0; cs1 *= f1;
#pragma unroll
            for (int d = 0; d < DP; ++d) { v0[d] *= f0; v1[d] *= f1; }
            // row sums of the wavefront's 16 rows -> LDS (lane 4 m holds the total of row m)
            {
                const double tot = wave_reduce16(racc);
                if ((lane & 3) == 0) s_rsum[lane >> 2] = tot;
                wave_lds_sync();
            }
            // lane partials of the tile's moments: column side for everybody, the row-side terms from the lanes < 16 (one row each)
            double pm[kTgMom];
#pragma unroll
            for (int k = 0; k < kTgMom; ++k) pm[k] = 0.0;
            pm[0] = cs0 + cs1;
            {
                // with y = V + (c / 2) u:  V + c u = y + (c / 2) u  and  c u u^T + V u^T + u V^T = y u^T + u y^T
                const double h0 = 0.5 * cs0, h1 = 0.5 * cs1;
                double y0[DP], y1[DP];
#pragma unroll
                for (int d = 0; d < DP; ++d) {
                    y0[d] = fma(h0, u0[d], v0[d]);
                    y1[d] = fma(h1, u1[d], v1[d]);
                    pm[1 + d] = fma(h0, u0[d], y0[d]) + fma(h1, u1[d], y1[d]);
                }
                int k = 0;
#pragma unroll
                for (int d = 0; d < DP; ++d)
#pragma unroll
                    for (int e = d; e < DP; ++e) {
                        pm[1 + DP + k] = fma(y0[d], u0[e], u0[d] * y0[e]) + fma(y1[d], u1[e], u1[d] * y1[e]);
                        ++k;
                    }
            }
            const double* xc = s_xs + (size_t)E * kTileW + 2 * lane;          // inputs of the lane's two columns
#pragma unroll
            for (int x = 0; x < NXC; ++x)
                if (x < NX) pm[1 + DP + NH + x] = cs0 * (xc[(D + x) * kTileW] - mo[D + x]) + cs1 * (xc[(D + x) * kTileW + 1] - mo[D + x]);
            if (lane < kTileRW) {
                const double rs = s_rsum[lane];
                const double* q = rec + lane * RSG;
                int k = 0;
#pragma unroll
                for (int d = 0; d < DP; ++d)
#pragma unroll
                    for (int e = d; e < DP; ++e) { pm[1 + DP + k] = fma(rs * q[1 + DP + d], q[1 + DP + e], pm[1 + DP + k]); ++k; }
                const double* xr = s_xs + wave * kTileRW + lane;                   // inputs of row (wave, lane)
#pragma unroll
                for (int x = 0; x < NXC; ++x)
                    if (x < NX) pm[1 + DP + NH + x] = fma(rs, xr[(D + x) * kTileW] - mo[D + x], pm[1 + DP + NH + x]);
            }
            wave_lds_sync();
            // the lane partials over the wavefront: 16 + 8 values
            {
                const double(&g16)[16] = *reinterpret_cast<const double(*)[16]>(&pm[0]);
                const double(&g8)[8] = *reinterpret_cast<const double(*)[8]>(&pm[16]);
                const double t16 = wave_reduce16(g16);
                const double t8 = wave_reduce8(g8);
                if ((lane & 3) == 0) {
                    wout[lane >> 2] = t16;
                    if (lane < 32) wout[16 + (lane >> 2)] = t8;
                }
            }
        }
    };

    // the wavefronts' sums of group g: fixed-order sum over the 8 wavefronts -> HBM (threads 0 .. 2 * kTgMom - 1)
    auto flush = [&](int g, const int (&Kg)[kTileGC]) {
        if (tid < kTileGC * kTgMom) {
            const int kk = tid / kTgMom, k = tid - kk * kTgMom;
            const int cl = g * kTileGC + kk;
            const int K = kk ? Kg[1] : Kg[0];
            if (cl < ncand) {
                double v = 0.0;
                if (K > 0 || (FUSED && k == 0)) {
                    const double* w = s_wsum + (size_t)(((g & 1) * kTileGC + kk) * kTileWaves) * kTgMom + k;
#pragma unroll
                    for (int wv = 0; wv < kTileWaves; ++wv) v += w[wv * kTgMom];
                }
                if (FUSED && k == 0) p.part[((size_t)(c0 + cl) * D + a) * p.ntiles + tile] = v;      // the forward's tile sum (step_combine_kernel)
                tmom[(((size_t)(c0 + cl) * D + a) * p.ntiles + tile) * kTgMom + k] = (K > 0) ? v : 0.0;
            }
        }
    };

    static_assert(kTileGC == 2 && kTgMom == 24, "two items per group; partial moments reduced as 16 + 8");
    int Kc[kTileGC], Kn[kTileGC];
    degrees(0, Kc);
    records(0, Kc);
    __syncthreads();
    for (int g = 0; g < ngroups; ++g) {
        degrees(g + 1, Kn);
        if (g + 1 < ngroups) records(g + 1, Kn);
        pairs(g, Kc);
        __syncthreads();
        flush(g, Kc);
#pragma unroll
        for (int kk = 0; kk < kTileGC; ++kk) Kc[kk] = Kn[kk];
    }
}

// Tiles of a diagonal pair in a fixed order, the factor 2 of the symmetric square, 1 / l_ax^2 on Pe -> the pair's slot of the
// moment array [W | P1 (DP) | P2 upper triangle | Pe (NXP)] and its flag for the element-wise kernels.
template <int DP>
__global__ __launch_bounds__(64) void tile_moments_reduce_kernel(const StepArgs p, const double* __restrict__ tmom, double* __restrict__ mom,
                                                                  int* __restrict__ done, int NSP, int NXP) {
    constexpr int NH = DP * (DP + 1) / 2;
    const int it = blockIdx.x, lane = threadIdx.x;          // item of this launch; p.item0 + it in the batch
    const int D = p.D, E = p.E, NX = E - D, P = D * (D + 1) / 2;
    const size_t git = p.fused_t >= 0 ? (size_t)it * p.H + p.fused_t : (size_t)p.item0 + it;      // FUSED: item = candidate, this step
    for (int a = 0; a < D; ++a) {
        const int q = pair_index(a, a, D);
        const int K = (int)p.crec[(size_t)it * p.CS + p.off_pair + (p.compact ? a : q) * p.PRP + DP * DP + 1] & 63;
        if (lane == 0) done[git * P + q] = K > 0 ? 1 : 0;
        if (K == 0) continue;
        if (lane < NSP) {
            // which partial moment feeds output slot `lane`
            int src = -1;
            double scale = 2.0;
            if (lane < 1 + DP + NH) src = lane;
            else if (lane - 1 - DP - NH < NX) {
                src = lane;
                scale = 2.0 * p.ils2[(size_t)a * E + D + (lane - 1 - DP - NH)];
            }
            double v = 0.0;
            if (src >= 0) {
                const double* tm = tmom + ((size_t)it * D + a) * p.ntiles * kTgMom + src;
                for (int k = 0; k < p.ntiles; ++k) v += tm[(size_t)k * kTgMom];
            }
            mom[(git * P + q) * NSP + lane] = v * scale;
        }
        (void)NXP;
    }
}

}  // namespace gpmpc_hip
