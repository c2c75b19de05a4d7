// pair_tile_kernel.h -- batch-major pairwise pass of the diagonal output pairs (gfx950 / MI355X).
//
// The N x N work of an output pair a == a (reference rl_gp_mpc/control_objects/models/gp_model.py:161-175:
// maha, L = exp(k_a + k_a' + maha), the two contractions with beta_a beta_a^T and iK_a) needs the table
// T_a = beta_a beta_a^T - iK_a (upper triangle, diagonal halved: rollout_kernel.h).  In the fused-horizon kernel a
// workgroup is a candidate and streams the whole table at every horizon step; once the D tables no longer stay in
// the 4 MiB L2 of an XCD (config 4: N = 1000, D = 4 -> 16 MB of upper triangles) every candidate re-reads them
// from the Infinity Cache / HBM: 674 GB per 2048-candidate batch, fabric-bound (round-2 counters).
//
// Here the roles are swapped for that part of a step: a workgroup OWNS one 128 x 128 tile of T_a in registers
// (8 wavefronts x 16 rows x 128 columns, lane = two adjacent columns) and loops over a chunk of candidates.  Per
// candidate it needs only the per-point factors of the tile's 128 row and 128 column points, which it recomputes
// from X (kept in LDS) and the candidate's D x D quantities (Z = R^-1 Sigma, the input mean, the Taylor degree:
// step_params_kernel) -- ~10 % more arithmetic, no table traffic: T is read once per (tile, chunk).
// One launch per horizon step (the step's state must exist); partial sums per (candidate, a, tile) go to HBM and
// are added in a fixed order by the per-candidate step kernel (rollout_kernel<.., TILED>), which keeps everything
// that is O(N) per candidate: mean part, separable off-diagonal pairs, the D x D update.
//
// Same element arithmetic as item_taylor2 / item_exp2 of rollout_kernel.h (Taylor polynomial of exp(g_i . w_j) of the
// degree the data range allows, table-based exp otherwise), fixed summation order (rows inside a wavefront's
// accumulators, DPP wave sum, the 8 wavefronts, then the tiles) => bitwise reproducible and independent of the batch.
#pragma once
#include "point_pass_kernel.h"

namespace gpmpc_hip {

constexpr int kTileW = 128;        // tile edge (rows and columns of T_a per workgroup)
constexpr int kTileRW = 16;        // rows per wavefront
constexpr int kTileWaves = 8;      // wavefronts per workgroup (kTileW / kTileRW)
constexpr int kTileGC = 2;         // candidates per barrier interval (records double-buffered)

// ------------------------------------------------------------------------------------------
// 16 rows x 2 columns per lane, Taylor form.  rec: LDS row records {ea_i, g_i[DP]} (stride RSR, broadcast reads);
// tv: the lane's T values; u0 / u1: w_j of the lane's two columns.  Two accumulators per column (rows alternate).
template <int DP, int K, int RSR>
__device__ inline void tile_rows_taylor(const double* rec, const double (&tv)[kTileRW][2], const double (&u0)[DP],
                                        const double (&u1)[DP], double& out0, double& out1) {
    double a00 = 0.0, a01 = 0.0, a10 = 0.0, a11 = 0.0;
#pragma unroll
    for (int r = 0; r < kTileRW; r += 2) {
        const double* q0 = rec + r * RSR;
        const double* q1 = q0 + RSR;
        const double e0 = q0[0], e1 = q1[0];
        double c00 = q0[1] * u0[0], c01 = q0[1] * u1[0], c10 = q1[1] * u0[0], c11 = q1[1] * u1[0];
#pragma unroll
        for (int d = 1; d < DP; ++d) {
            c00 = fma(q0[1 + d], u0[d], c00);
            c01 = fma(q0[1 + d], u1[d], c01);
            c10 = fma(q1[1 + d], u0[d], c10);
            c11 = fma(q1[1 + d], u1[d], c11);
        }
        a00 = fma(taylor_exp<K>(c00) * e0, tv[r][0], a00);
        a01 = fma(taylor_exp<K>(c01) * e0, tv[r][1], a01);
        a10 = fma(taylor_exp<K>(c10) * e1, tv[r + 1][0], a10);
        a11 = fma(taylor_exp<K>(c11) * e1, tv[r + 1][1], a11);
    }
    out0 = a00 + a10;
    out1 = a01 + a11;
}

// Direct form exp(ka'_i + kb'_j + g_i . w_j): records hold ka'_i, k0 / k1 are kb' of the lane's columns.
template <int DP, int RSR>
__device__ inline void tile_rows_exp(const double* rec, const double (&tv)[kTileRW][2], const double (&u0)[DP],
                                     const double (&u1)[DP], double k0, double k1, const double* tab, double& out0, double& out1) {
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int r = 0; r < kTileRW; ++r) {
        const double* q = rec + r * RSR;
        double x = q[0] + k0, y = q[0] + k1;
#pragma unroll
        for (int d = 0; d < DP; ++d) { x = fma(q[1 + d], u0[d], x); y = fma(q[1 + d], u1[d], y); }
        a0 = fma(fast_exp(x, tab), tv[r][0], a0);
        a1 = fma(fast_exp(y, tab), tv[r][1], a1);
    }
    out0 = a0;
    out1 = a1;
}

struct TileLayout {
    int xs, rows, cols, wsum, tab, total;     // offsets in doubles
};

__host__ __device__ inline TileLayout make_tile_layout(int DP, int E, int cch) {
    const int RSR = (DP + 2) & ~1;
    TileLayout L;
    int o = 0;
    L.tab = o;  o += 64;
    L.xs = o;   o += 2 * E * kTileW;                          // [side][e][point]
    L.rows = o; o += 2 * kTileGC * kTileW * RSR;              // [buffer][candidate of the group][row][RSR]
    L.cols = o; o += 2 * kTileGC * (DP + 1) * kTileW;         // [buffer][candidate][component][column]
    L.wsum = o; o += ((cch + 1) & ~1) * kTileWaves;           // [candidate of the chunk][wavefront]
    L.total = o;
    return L;
}

// ------------------------------------------------------------------------------------------
template <int DP>
__global__ __launch_bounds__(kTileWaves * 64, 4) void pair_tile_kernel(const StepArgs p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int RSR = (DP + 2) & ~1;
    constexpr int NTH = kTileWaves * 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = p.N, D = p.D, E = p.E;

    // block -> (output a, tile, chunk of candidates).  All chunks of one (a, tile) go to the same XCD (blocks b, b + 8,
    // ... share an L2): the tile's 128 KiB are fetched from the fabric once per XCD and step, not once per chunk.
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

    const TileLayout L = make_tile_layout(DP, E, p.cch);
    double* s_tab = smem + L.tab;
    double* s_xs = smem + L.xs;
    double* s_rows = smem + L.rows;
    double* s_cols = smem + L.cols;
    double* s_wsum = smem + L.wsum;

    // ---- the tile of T_a: 16 rows x 2 adjacent columns per lane, in registers for the whole chunk -----------------
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
    // inputs of the tile's row and column points (points past N: the last point, their T entries are zero)
    for (int k = tid; k < 2 * E * kTileW; k += NTH) {
        const int side = k / (E * kTileW);
        const int rem = k - side * (E * kTileW);
        const int e = rem / kTileW, pt = rem - e * kTileW;
        int gp = (side ? j0 : i0) + pt;
        gp = gp < N ? gp : N - 1;
        s_xs[k] = p.Xt[(size_t)e * N + gp];
    }
    __syncthreads();

    const double* il = p.ils2 + (size_t)a * E;     // wave-uniform: scalar loads
    const double lv = p.logvar[a];
    const int ngroups = (ncand + kTileGC - 1) / kTileGC;
    // the (a, a) problem of the candidates' step records: Z (DP x DP) | 1 / sqrt(det R) | degree
    const double* __restrict__ tpar = p.crec + p.off_pair + pair_index(a, a, D) * p.PRP;
    // Taylor degrees of the candidates of a group (0: direct exp), fetched one group ahead
    auto degrees = [&](int g, int (&K)[kTileGC]) {
#pragma unroll
        for (int kk = 0; kk < kTileGC; ++kk) {
            const int cl = g * kTileGC + kk;
            K[kk] = (cl < ncand) ? ((int)tpar[(size_t)(c0 + cl) * p.CS + DP * DP + 1] & 63) : 0;
        }
    };

    // phase A: per-point factors of candidate group g into buffer (g & 1).  Wavefront w: candidate (w >> 2) of the group,
    // side (w >> 1) & 1 (0 rows, 1 columns), points (w & 1) * 64 + lane.
    auto records = [&](int g, const int (&Kg)[kTileGC]) {
        const int kk = wave >> 2;
        const int cl = g * kTileGC + kk;
        if (cl >= ncand) return;
        const double* par = tpar + (size_t)(c0 + cl) * p.CS;                  // wave-uniform
        const int K = kk ? Kg[1] : Kg[0];
        const double* mo = p.crec + (size_t)(c0 + cl) * p.CS;                  // input mean of the step
        const int side = (wave >> 1) & 1;
        const int pt = (wave & 1) * 64 + lane;
        const double* xp = s_xs + (size_t)side * E * kTileW + pt;
        // log-factor and monomial variables of the point from the pair's Q and G (step_params_kernel)
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
        const double f = (K > 0) ? fast_exp(kkv, s_tab) : kkv;
        const int buf = g & 1;
        if (side == 0) {
            double* rec = s_rows + ((size_t)(buf * kTileGC + kk) * kTileW + pt) * RSR;
            rec[0] = f;
#pragma unroll
            for (int d = 0; d < DP; ++d) rec[1 + d] = g_[d];
        } else {
            double* col = s_cols + (size_t)(buf * kTileGC + kk) * (DP + 1) * kTileW + pt;
            col[0] = f;
#pragma unroll
            for (int d = 0; d < DP; ++d) col[(1 + d) * kTileW] = u[d];
        }
    };

    // phase B: the tile against the records of group g
    auto pairs = [&](int g, const int (&Kg)[kTileGC]) {
        const int buf = g & 1;
#pragma unroll 1
        for (int kk = 0; kk < kTileGC; ++kk) {
            const int cl = g * kTileGC + kk;
            if (cl >= ncand) break;
            const int K = kk ? Kg[1] : Kg[0];
            const double* col = s_cols + (size_t)(buf * kTileGC + kk) * (DP + 1) * kTileW + 2 * lane;
            const double f0 = col[0], f1 = col[1];
            double u0[DP], u1[DP];
#pragma unroll
            for (int d = 0; d < DP; ++d) { u0[d] = col[(1 + d) * kTileW]; u1[d] = col[(1 + d) * kTileW + 1]; }
            const double* rec = s_rows + ((size_t)(buf * kTileGC + kk) * kTileW + wave * kTileRW) * RSR;
            double s0, s1, v;
            if (K == 0) {
                tile_rows_exp<DP, RSR>(rec, tv, u0, u1, f0, f1, s_tab, s0, s1);
                v = s0 + s1;
            } else {
                if (K <= 2) tile_rows_taylor<DP, 2, RSR>(rec, tv, u0, u1, s0, s1);
                else if (K == 3) tile_rows_taylor<DP, 3, RSR>(rec, tv, u0, u1, s0, s1);
                else if (K == 4) tile_rows_taylor<DP, 4, RSR>(rec, tv, u0, u1, s0, s1);
                else if (K == 5) tile_rows_taylor<DP, 5, RSR>(rec, tv, u0, u1, s0, s1);
                else if (K == 6) tile_rows_taylor<DP, 6, RSR>(rec, tv, u0, u1, s0, s1);
                else if (K <= 8) tile_rows_taylor<DP, 8, RSR>(rec, tv, u0, u1, s0, s1);
                else if (K <= 10) tile_rows_taylor<DP, 10, RSR>(rec, tv, u0, u1, s0, s1);
                else if (K <= 12) tile_rows_taylor<DP, 12, RSR>(rec, tv, u0, u1, s0, s1);
                else tile_rows_taylor<DP, 14, RSR>(rec, tv, u0, u1, s0, s1);
                v = fma(s0, f0, s1 * f1);
            }
            v = wave_sum(v);
            if (lane == 0) s_wsum[cl * kTileWaves + wave] = v;
        }
    };

    // One barrier per group: records of group g + 1 are written (other buffer) before the pairs of group g are consumed;
    // a wavefront reaches the barrier after records(g + 1) only once it is done with pairs(g - 1), so buffer g & 1 is free
    // when records(g + 2) starts.
    static_assert(kTileGC == 2, "the degree selects above assume two candidates per group");
    int Kc[kTileGC], Kn[kTileGC];
    degrees(0, Kc);
    records(0, Kc);
    __syncthreads();
    for (int g = 0; g < ngroups; ++g) {
        degrees(g + 1, Kn);
        if (g + 1 < ngroups) records(g + 1, Kn);
        pairs(g, Kc);
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < kTileGC; ++kk) Kc[kk] = Kn[kk];
    }
    // fixed-order sum over the wavefronts; the factor 2 (i <= j only) and 1 / sqrt(det R) are applied by the step kernel
    for (int cl = tid; cl < ncand; cl += NTH) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < kTileWaves; ++w) v += s_wsum[cl * kTileWaves + w];
        p.part[((size_t)(c0 + cl) * D + a) * p.ntiles + tile] = v;
    }
}

}  // namespace gpmpc_hip
