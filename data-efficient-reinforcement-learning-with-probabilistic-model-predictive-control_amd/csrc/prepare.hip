// prepare.hip -- once-per-control-step factorisation for gfx950 (MI355X, CDNA4).
//
// Replaces calculate_factorizations + prepare_inference of the reference
// (rl_gp_mpc/control_objects/models/gp_model.py:400-431, 182-191):
//   K_a   = outputscale_a * exp(-1/2 sum_e ((x_ie - x_je)/l_ae)^2) + noise_a I     (:425,427)
//   L_a   = chol(K_a)                       blocked right-looking, fp64 MFMA trailing update (:427)
//   Y_a   = L_a^-1                          blocked forward substitution, fp64 MFMA
//   beta  = Y^T (Y y)                       (= cholesky_solve(y, L), :429-430)
//   iK_a  = Y_a^T Y_a                       fp64 MFMA (= cholesky_solve(I, L), :428)
//   T_a   = beta_a beta_a^T - iK_a, diagonal halved   (table streamed by the rollout kernel)
// The Gram build is HBM-write-bound (one coalesced 8-byte store per element); the three
// N^3 contractions are the only dense contractions of the hot path and run on the matrix
// cores with v_mfma_f64_16x16x4_f64 (wave64: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15],
// D[row=(l>>4)+4r][col=l&15]).
#include "gpmpc_internal.h"
#include "prepare_tiled.h"
#include "rollout_kernel.h"        // fast_exp / kExp2Tab

namespace gpmpc_hip {

// acc += sum over p in [pbeg, pend) of A(p) B(p) for one 16 x 16 tile with the operands straight from global memory:
// the loads of U k-steps are issued before the first MFMA of the group, so a group costs one memory round trip, not U
// (without it every v_mfma waited for its own two loads: the N^3 kernels ran at L2 latency, 8-21 TFLOP/s at N = 4096).
template <int U, typename FA, typename FB>
__device__ inline void mfma_kloop(d4& acc, int pbeg, int pend, int lk, FA loadA, FB loadB) {
    for (int pp = pbeg; pp < pend; pp += 4 * U) {
        double av[U], bv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int pk = pp + 4 * u + lk;
            const bool in = pk < pend;
            av[u] = in ? loadA(pk) : 0.0;
            bv[u] = in ? loadB(pk) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
    }
}

// ------------------------------------------------------------------------------------------
__global__ void pack_inputs_kernel(const double* __restrict__ X, const double* __restrict__ ls,
                                   const double* __restrict__ os, int N, int D, int E,
                                   double* __restrict__ Xt, double* __restrict__ ils2,
                                   double* __restrict__ var, double* __restrict__ logvar) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < N * E) {
        const int e = idx / N, pt = idx - e * N;
        Xt[idx] = X[(size_t)pt * E + e];
    }
    if (idx < D * E) { const double l = ls[idx]; ils2[idx] = 1.0 / (l * l); }
    if (idx < D) { var[idx] = os[idx]; logvar[idx] = log(os[idx]); }
}

// per-input-dimension min / max over the memory points (bounds |x - m| in the rollout kernel)
__global__ __launch_bounds__(256) void xrange_kernel(const double* __restrict__ X, int N, int E, double* __restrict__ xr) {
    __shared__ double smin[4], smax[4];
    const int e = blockIdx.x;
    double lo = INFINITY, hi = -INFINITY;
    for (int pt = threadIdx.x; pt < N; pt += 256) {
        const double v = X[(size_t)pt * E + e];
        lo = fmin(lo, v);
        hi = fmax(hi, v);
    }
    for (int off = 32; off >= 1; off >>= 1) {
        lo = fmin(lo, __shfl_xor(lo, off, 64));
        hi = fmax(hi, __shfl_xor(hi, off, 64));
    }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        xr[e] = fmin(fmin(smin[0], smin[1]), fmin(smin[2], smin[3]));
        xr[E + e] = fmax(fmax(smax[0], smax[1]), fmax(smax[2], smax[3]));
    }
}

// pack_inputs + xrange + the record of (X, Y, hyper-parameters) the factors are computed from, in ONE launch (they were
// 2 kernels + 5 device-to-device copies: at small N every launch is ~5 us of a 0.1-0.2 ms call).  Blocks e < E also reduce
// the data range of input dimension e.
__global__ __launch_bounds__(256) void pack_record_kernel(const double* __restrict__ X, const double* __restrict__ Y,
                                                          const double* __restrict__ ls, const double* __restrict__ os,
                                                          const double* __restrict__ noise, int N, int D, int E,
                                                          double* __restrict__ Xt, double* __restrict__ ils2, double* __restrict__ var,
                                                          double* __restrict__ logvar, double* __restrict__ xr,
                                                          double* __restrict__ Xc, double* __restrict__ Yc, double* __restrict__ hyp) {
    const int stride = gridDim.x * 256;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < N * E; idx += stride) {
        const int e = idx / N, pt = idx - e * N;
        Xt[idx] = X[(size_t)pt * E + e];
        Xc[idx] = X[idx];
    }
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < N * D; idx += stride) Yc[idx] = Y[idx];
    if (blockIdx.x == 0) {
        for (int idx = threadIdx.x; idx < D * E; idx += 256) { const double l = ls[idx]; ils2[idx] = 1.0 / (l * l); hyp[idx] = l; }
        if (threadIdx.x < D) {
            var[threadIdx.x] = os[threadIdx.x]; logvar[threadIdx.x] = log(os[threadIdx.x]);
            hyp[(size_t)D * E + threadIdx.x] = os[threadIdx.x];
            hyp[(size_t)D * E + kMaxD + threadIdx.x] = noise[threadIdx.x];
        }
    }
    if ((int)blockIdx.x < E) {
        __shared__ double smin[4], smax[4];
        const int e = blockIdx.x;
        double lo = INFINITY, hi = -INFINITY;
        for (int pt = threadIdx.x; pt < N; pt += 256) { const double v = X[(size_t)pt * E + e]; lo = fmin(lo, v); hi = fmax(hi, v); }
        for (int off = 32; off >= 1; off >>= 1) { lo = fmin(lo, __shfl_xor(lo, off, 64)); hi = fmax(hi, __shfl_xor(hi, off, 64)); }
        if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            xr[e] = fmin(fmin(smin[0], smin[1]), fmin(smin[2], smin[3]));
            xr[E + e] = fmax(fmax(smax[0], smax[1]), fmax(smax[2], smax[3]));
        }
    }
}

// Is the cached (X, Y, hyper-parameters) a prefix of the new one?  One launch over the five arrays; `flag` is zero when
// idle (the host clears it again after a mismatch).
__global__ __launch_bounds__(256) void prefix_mismatch_all_kernel(const double* __restrict__ X, const double* __restrict__ Xc, size_t nX,
                                                                  const double* __restrict__ Y, const double* __restrict__ Yc, size_t nY,
                                                                  const double* __restrict__ ls, const double* __restrict__ os,
                                                                  const double* __restrict__ noise, const double* __restrict__ hyp,
                                                                  int D, int E, int* __restrict__ flag) {
    const size_t stride = (size_t)gridDim.x * 256;
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nX; i += stride) bad |= (X[i] != Xc[i]);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nY; i += stride) bad |= (Y[i] != Yc[i]);
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < D * E; i += 256) bad |= (ls[i] != hyp[i]);
        if ((int)threadIdx.x < D) bad |= (os[threadIdx.x] != hyp[(size_t)D * E + threadIdx.x]) || (noise[threadIdx.x] != hyp[(size_t)D * E + kMaxD + threadIdx.x]);
    }
    if (bad) *flag = 1;
}

// K_a = outputscale_a exp(-1/2 sum_e ((x_ie - x_je)/l_ae)^2) + noise_a I, 64 x 64 tiles of the upper block
// triangle (tj >= ti); the mirror tile is written through an LDS transpose, so every store is a coalesced
// 512-byte row segment and every element is computed once.  Lane = column j (its x_j / l_a in registers),
// the 16 rows a thread visits read x_i / l_a as LDS broadcasts: per element 2E VALU + one exp and no global
// load, which puts the kernel on the HBM-write side of its roofline (D N^2 8 bytes out).
template <int EP>
__global__ __launch_bounds__(256) void gram_kernel(const double* __restrict__ Xt, const double* __restrict__ ils2,
                                                   const double* __restrict__ var, const double* __restrict__ noise,
                                                   int N, int E, double* __restrict__ K, int lower_only) {
    __shared__ double xi[64][EP + 1];          // rows of the tile, pre-scaled by 1 / l_a
    __shared__ double tile[64][65];
    const int a = blockIdx.z;
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (tj < ti) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = ti * 64, j0 = tj * 64;
    double il[EP];
#pragma unroll
    for (int e = 0; e < EP; ++e) il[e] = (e < E) ? sqrt(ils2[a * E + e]) : 0.0;
    for (int idx = threadIdx.x; idx < 64 * EP; idx += 256) {
        const int r = idx / EP, e = idx - r * EP;
        const int i = i0 + r;
        xi[r][e] = (e < E && i < N) ? Xt[(size_t)e * N + i] * sqrt(ils2[a * E + e]) : 0.0;
    }
    const int j = j0 + lane;
    double xj[EP];
#pragma unroll
    for (int e = 0; e < EP; ++e) xj[e] = (e < E && j < N) ? Xt[(size_t)e * N + j] * il[e] : 0.0;
    __syncthreads();
    const double va = var[a], nz = noise[a];
    double* Ka = K + (size_t)a * N * N;
#pragma unroll 4
    for (int rr = 0; rr < 16; ++rr) {
        const int r = wave * 16 + rr;
        double s = 0.0;
#pragma unroll
        for (int e = 0; e < EP; ++e) { const double d = xi[r][e] - xj[e]; s = fma(d, d, s); }
        double v = va * exp(-0.5 * s);
        const int i = i0 + r;
        if (i == j) v += nz;
        tile[r][lane] = v;
        // lower_only: the factorisation reads the lower triangle only -- the upper off-diagonal tiles are not stored
        // (half of the kernel's HBM writes at large N)
        if (i < N && j < N && (tj == ti || !lower_only)) Ka[(size_t)i * N + j] = v;
    }
    if (tj != ti) {                                  // mirror tile: K[j0 + r][i0 + lane] = tile[lane][r]
        __syncthreads();
#pragma unroll 4
        for (int rr = 0; rr < 16; ++rr) {
            const int r = wave * 16 + rr;
            const int jj = j0 + r, ii = i0 + lane;
            if (jj < N && ii < N) Ka[(size_t)jj * N + ii] = tile[lane][r];
        }
    }
}

// The same matrices for large memories (lower triangle only, N >= outer_min_n), all D GPs by ONE workgroup per 64 x 64 tile:
// the squared differences (x_ie - x_je)^2 of an element are formed once and contracted with every GP's 1 / l_ae^2 (LDS
// broadcasts), instead of being recomputed per GP -- per stored element E FMAs + a table-based exp (fast_exp, ~1 ulp) where
// gram_kernel spends 2 EP VALU + libm's exp (139 vector instructions per element at config 5: it ran at 1.05 TB/s of stores,
// VALU-bound).  Lane = column j (coalesced 512-byte row segments per GP), a thread visits its 16 rows two at a
// time (one set of LDS reads of 1 / l^2 serves both).  Tiles with tj > ti are not launched at all (grid over the lower block
// triangle through a linear tile index).
template <int EP>
__global__ __launch_bounds__(256) void gram_lower_kernel(const double* __restrict__ Xt, const double* __restrict__ ils2,
                                                         const double* __restrict__ var, const double* __restrict__ noise,
                                                         int N, int E, int D, double* __restrict__ K) {
    __shared__ double xi[64][EP + 1];           // rows of the tile (unscaled)
    __shared__ __attribute__((aligned(16))) double s_il[kMaxD][EP];
    __shared__ double s_var[kMaxD], s_nz[kMaxD], s_tab[64];
    // linear index over the lower block triangle: t = ti (ti + 1) / 2 + tj, tj <= ti
    const int t = blockIdx.x;
    int ti = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    while (ti * (ti + 1) / 2 > t) --ti;
    const int tj = t - ti * (ti + 1) / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = ti * 64, j0 = tj * 64;
    for (int idx = threadIdx.x; idx < 64 * EP; idx += 256) {
        const int r = idx / EP, e = idx - r * EP;
        xi[r][e] = (e < E && i0 + r < N) ? Xt[(size_t)e * N + i0 + r] : 0.0;
    }
    for (int idx = threadIdx.x; idx < D * EP; idx += 256) {
        const int a = idx / EP, e = idx - a * EP;
        s_il[a][e] = (e < E) ? ils2[a * E + e] : 0.0;
    }
    if ((int)threadIdx.x < D) { s_var[threadIdx.x] = var[threadIdx.x]; s_nz[threadIdx.x] = noise[threadIdx.x]; }
    if (threadIdx.x >= 64 && threadIdx.x < 128) s_tab[threadIdx.x - 64] = kExp2Tab[threadIdx.x - 64];
    const int j = j0 + lane;
    double xj[EP];
#pragma unroll
    for (int e = 0; e < EP; ++e) xj[e] = (e < E && j < N) ? Xt[(size_t)e * N + j] : 0.0;
    __syncthreads();
    for (int rr = 0; rr < 16; rr += 2) {
        const int r0 = wave * 16 + rr, r1 = r0 + 1;
        double d0[EP], d1[EP];
#pragma unroll
        for (int e = 0; e < EP; ++e) {
            const double a0 = xi[r0][e] - xj[e], a1 = xi[r1][e] - xj[e];
            d0[e] = a0 * a0;
            d1[e] = a1 * a1;
        }
        const int ia = i0 + r0, ib = i0 + r1;
        for (int a = 0; a < D; ++a) {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int e = 0; e < EP; ++e) { const double w = s_il[a][e]; s0 = fma(w, d0[e], s0); s1 = fma(w, d1[e], s1); }
            double v0 = s_var[a] * fast_exp(-0.5 * s0, s_tab), v1 = s_var[a] * fast_exp(-0.5 * s1, s_tab);
            if (ia == j) v0 += s_nz[a];
            if (ib == j) v1 += s_nz[a];
            double* Ka = K + (size_t)a * N * N;
            if (j < N) {
                if (ia < N) Ka[(size_t)ia * N + j] = v0;
                if (ib < N) Ka[(size_t)ib * N + j] = v1;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Unblocked Cholesky of the nb x nb diagonal block at (k0, k0) and its inverse (for the
// triangular-inverse recursion).  One workgroup of NB*NB threads per GP.
__global__ __launch_bounds__(NB * NB) void potrf_diag_kernel(double* __restrict__ Kall, double* __restrict__ Yall,
                                                            int N, int k0, int nb, int* __restrict__ info) {
    __shared__ double s[NB][NB + 1];
    __shared__ double y[NB][NB + 1];
    const int a = blockIdx.x;
    double* K = Kall + (size_t)a * N * N;
    double* Y = Yall + (size_t)a * N * N;
    const int r = threadIdx.x / NB, c = threadIdx.x % NB;
    const bool in = (r < nb && c < nb);
    s[r][c] = (in && c <= r) ? K[(size_t)(k0 + r) * N + (k0 + c)] : 0.0;
    __syncthreads();
    for (int k = 0; k < nb; ++k) {
        if (r == k && c == k) {
            const double d = s[k][k];
            if (!(d > 0.0) && info[a] == 0) info[a] = k0 + k + 1;
            s[k][k] = sqrt(d);
        }
        __syncthreads();
        if (c == k && r > k && r < nb) s[r][k] /= s[k][k];
        __syncthreads();
        if (in && c > k && c <= r) s[r][c] -= s[r][k] * s[c][k];
        __syncthreads();
    }
    if (in && c <= r) K[(size_t)(k0 + r) * N + (k0 + c)] = s[r][c];
    // inverse of the lower-triangular block by forward substitution on the identity, one row of Y per
    // step, the whole nb x nb thread grid applying the rank-1 update (depth nb instead of nb^2 / 2)
    y[r][c] = (r == c) ? 1.0 : 0.0;
    __syncthreads();
    for (int k = 0; k < nb; ++k) {
        if (r == k && c <= k) y[k][c] /= s[k][k];
        __syncthreads();
        if (in && r > k && c <= k) y[r][c] -= s[r][k] * y[k][c];
        __syncthreads();
    }
    if (in) Y[(size_t)(k0 + r) * N + (k0 + c)] = (c <= r) ? y[r][c] : 0.0;
}

// Panel solve: rows below the diagonal block, A21 <- A21 L11^-T  (one thread per row).
__global__ __launch_bounds__(256) void trsm_panel_kernel(double* __restrict__ Kall, int N, int k0, int nb) {
    __shared__ double l11[NB][NB + 1];
    const int a = blockIdx.y;
    double* K = Kall + (size_t)a * N * N;
    for (int idx = threadIdx.x; idx < NB * NB; idx += blockDim.x) {
        const int r = idx / NB, c = idx % NB;
        l11[r][c] = (r < nb && c <= r) ? K[(size_t)(k0 + r) * N + (k0 + c)] : 0.0;
    }
    __syncthreads();
    const int row = k0 + nb + blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= N) return;
    double x[NB];
    double* rp = K + (size_t)row * N + k0;
#pragma unroll
    for (int c = 0; c < NB; ++c) x[c] = (c < nb) ? rp[c] : 0.0;
#pragma unroll
    for (int c = 0; c < NB; ++c) {
        if (c < nb) {
            double v = x[c];
#pragma unroll
            for (int m = 0; m < NB; ++m)
                if (m < c) v = fma(-x[m], l11[c][m], v);
            x[c] = v / l11[c][c];
        }
    }
#pragma unroll
    for (int c = 0; c < NB; ++c)
        if (c < nb) rp[c] = x[c];
}

// Diagonal block of a panel, round 2: the same arrangement as prepare_small.hip's pivot loop.  Thread (row rr, column c)
// owns element (rr, c) of the 32 x 32 block and of the 32 x 32 identity under it; 1 / sqrt(pivot) from an fp32 v_rsq seed +
// two Newton steps (6 dependent fp64 operations instead of ~35 for sqrt + divide -- a dependent v_fma_f64 issues only every
// ~40 cycles on this part), computed by the owner of the NEXT pivot right after its own update: one barrier per pivot;
// columns are never divided, L is scaled on the way out; the identity comes out as L11^-T, i.e. the block's inverse that
// the panel solve and the triangular inverse use.  25 -> ~9 us per panel.

// left > 0 (outer-blocked path): the block first receives the update of the `left` columns before it inside the current
// outer panel, A_kk -= L[k, J:k0] L[k, J:k0]^T -- the panels of an outer panel are factorised left-looking, so no
// trailing update is launched (and no trailing strip re-read and re-written) per 32 columns.
__global__ __launch_bounds__(NB * NB) void potrf_diag_fast_kernel(double* __restrict__ Kall, double* __restrict__ Yall,
                                                                 int N, int k0, int nb, int* __restrict__ info, int left) {
    __shared__ double colb[2][2 * NB];
    __shared__ double sinv[NB];
    __shared__ double lb[NB][3 * NB + 1];
    const int a = blockIdx.x;
    double* K = Kall + (size_t)a * N * N;
    double* Y = Yall + (size_t)a * N * N;
    const int tid = threadIdx.x, wave = tid >> 6;
    const int c = tid >> 5, r = tid & 31;                    // column-major over the wavefronts: wave w <-> columns 2w, 2w + 1
    double a0 = (r < nb && c < nb && c <= r) ? K[(size_t)(k0 + r) * N + (k0 + c)] : 0.0;
    if (left > 0) {
        for (int idx = tid; idx < NB * left; idx += NB * NB) {
            const int rr = idx / left, p = idx - rr * left;
            lb[rr][p] = (rr < nb) ? K[(size_t)(k0 + rr) * N + (k0 - left + p)] : 0.0;
        }
        __syncthreads();
        double u0 = 0.0, u1 = 0.0, u2 = 0.0, u3 = 0.0;
        for (int p = 0; p < left; p += 4) {
            u0 = fma(lb[r][p], lb[c][p], u0);
            u1 = fma(lb[r][p + 1], lb[c][p + 1], u1);
            u2 = fma(lb[r][p + 2], lb[c][p + 2], u2);
            u3 = fma(lb[r][p + 3], lb[c][p + 3], u3);
        }
        if (r < nb && c < nb && c <= r) a0 -= (u0 + u1) + (u2 + u3);
    }
    double a1 = (r == c) ? 1.0 : 0.0;
    if (c == 0) { colb[0][r] = a0; colb[0][NB + r] = a1; }
    if (tid == 0) {
        if (!(a0 > 0.0) && info[a] == 0) info[a] = k0 + 1;
        sinv[0] = inv_sqrt_pos_p(a0);
    }
    if (tid >= nb && tid < NB) sinv[tid] = 0.0;
    __syncthreads();
    for (int k = 0; k + 1 < nb; ++k) {
        if (2 * wave + 1 > k) {                              // wave-uniform: both columns of a finished wave are final
            const double* cb = colb[k & 1];
            double* cn = colb[(k + 1) & 1];
            const double inv = sinv[k];
            if (c > k && c < nb) {
                const double lc = cb[c] * inv;
                a0 = fma(-(cb[r] * inv), lc, a0);
                a1 = fma(-(cb[NB + r] * inv), lc, a1);
                if (c == k + 1) {
                    cn[r] = a0;
                    cn[NB + r] = a1;
                    if (r == k + 1) {
                        if (!(a0 > 0.0) && info[a] == 0) info[a] = k0 + k + 2;
                        sinv[k + 1] = inv_sqrt_pos_p(a0);
                    }
                }
            }
        }
        __syncthreads();
    }
    const double sc = sinv[c];                               // 0 for c >= nb
    if (r < nb && c <= r) K[(size_t)(k0 + r) * N + k0 + c] = a0 * sc;
    if (r < nb && c < nb) Y[(size_t)(k0 + c) * N + k0 + r] = (r <= c) ? a1 * sc : 0.0;       // row r of the identity -> column r of Y11
}

// Panel solve on the matrix cores: L21 = A21 Y11^T (Y11 = the block's inverse written by potrf_diag_fast_kernel).
// One workgroup = 64 rows (a 16-row tile per wavefront, both 16-column halves); memory-bound: M x 32 in, M x 32 out.
__global__ __launch_bounds__(256) void trsm_panel_mfma_kernel(double* __restrict__ Kall, const double* __restrict__ Yall,
                                                              int N, int k0, int nb) {
    __shared__ double yt[NB][NB + 1];                        // yt[k][j] = Y11[j][k]
    const int a = blockIdx.y;
    double* K = Kall + (size_t)a * N * N;
    const double* Y = Yall + (size_t)a * N * N;
    {
        // the four loads of a thread in flight (clamped address + select; a branch per element makes the compiler wait for
        // every load before the next: 4 L2 round trips on the critical path of every panel)
        const int k = threadIdx.x & 31, j0 = threadIdx.x >> 5;
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 8 * u;
            const int jj = j < nb ? j : nb - 1, kk = k <= jj ? k : jj;
            v[u] = Y[(size_t)(k0 + jj) * N + k0 + kk];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 8 * u;
            yt[k][j] = (j < nb && k <= j) ? v[u] : 0.0;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int i0 = k0 + nb + blockIdx.x * 64 + wave * 16;
    if (i0 >= N) return;
    const int ri = i0 + li;
    const double* Ar = K + (size_t)(ri < N ? ri : N - 1) * N + k0;
    double av[NB / 4];
#pragma unroll
    for (int q = 0; q < NB / 4; ++q) av[q] = (ri < N && 4 * q + lk < nb) ? Ar[4 * q + lk] : 0.0;
    d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < NB / 4; ++q) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], yt[4 * q + lk][li], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], yt[4 * q + lk][16 + li], acc1, 0, 0, 0);
    }
    // every lane of the wavefront has read its A values before any lane stores (same wavefront: program order)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = i0 + lk + 4 * r;
        if (row < N) {
            if (li < nb) K[(size_t)row * N + k0 + li] = acc0[r];
            if (16 + li < nb) K[(size_t)row * N + k0 + 16 + li] = acc1[r];
        }
    }
}

// Panel solve of the left-looking inner level: rows below the diagonal block first receive the update of the `left`
// columns before the panel inside the current outer panel, X = A[rows, k0:k0+32] - L[rows, J:k0] L[k0:k0+32, J:k0]^T, then
// L21 = X Y11^T.  One workgroup = 64 rows; the update leaves the accumulators in the C layout, the solve wants X as the
// A operand: one trip through LDS per wavefront.
__global__ __launch_bounds__(256) void trsm_panel_ll_kernel(double* __restrict__ Kall, const double* __restrict__ Yall,
                                                            int N, int k0, int nb, int left) {
    __shared__ double yt[NB][NB + 1];                        // yt[k][j] = Y11[j][k]
    __shared__ double lbt[3 * NB][NB + 1];                   // lbt[p][j] = L[k0 + j][k0 - left + p]
    __shared__ double xs[4][16][NB + 1];
    const int a = blockIdx.y;
    double* K = Kall + (size_t)a * N * N;
    const double* Y = Yall + (size_t)a * N * N;
    for (int idx = threadIdx.x; idx < NB * NB; idx += 256) {
        const int j = idx / NB, k = idx % NB;
        yt[k][j] = (j < nb && k <= j) ? Y[(size_t)(k0 + j) * N + k0 + k] : 0.0;
    }
    for (int idx = threadIdx.x; idx < NB * left; idx += 256) {
        const int j = idx / left, p = idx - j * left;
        lbt[p][j] = (j < nb) ? K[(size_t)(k0 + j) * N + (k0 - left + p)] : 0.0;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int i0 = k0 + nb + blockIdx.x * 64 + wave * 16;    // wavefronts past the last row run along (loads masked, no stores)
    const int ri = i0 + li;
    const double* Ar = K + (size_t)(ri < N ? ri : N - 1) * N + (k0 - left);
    d4 u0 = {0.0, 0.0, 0.0, 0.0}, u1 = {0.0, 0.0, 0.0, 0.0};
    double cold0[4], cold1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {                            // the old values travel while the update is formed
        const int row = i0 + lk + 4 * r;
        cold0[r] = (row < N && li < nb) ? K[(size_t)row * N + k0 + li] : 0.0;
        cold1[r] = (row < N && 16 + li < nb) ? K[(size_t)row * N + k0 + 16 + li] : 0.0;
    }
    for (int p = 0; p < left; p += 16) {
        double al[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) al[q] = (ri < N) ? Ar[p + 4 * q + lk] : 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u0 = __builtin_amdgcn_mfma_f64_16x16x4f64(al[q], lbt[p + 4 * q + lk][li], u0, 0, 0, 0);
            u1 = __builtin_amdgcn_mfma_f64_16x16x4f64(al[q], lbt[p + 4 * q + lk][16 + li], u1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        xs[wave][lk + 4 * r][li] = cold0[r] - u0[r];
        xs[wave][lk + 4 * r][16 + li] = cold1[r] - u1[r];
    }
    __syncthreads();                                         // lanes exchange through xs: needs the barrier's ordering
    double av[NB / 4];
#pragma unroll
    for (int q = 0; q < NB / 4; ++q) av[q] = xs[wave][li][4 * q + lk];
    d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < NB / 4; ++q) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], yt[4 * q + lk][li], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], yt[4 * q + lk][16 + li], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = i0 + lk + 4 * r;
        if (row < N) {
            if (li < nb) K[(size_t)row * N + k0 + li] = acc0[r];
            if (16 + li < nb) K[(size_t)row * N + k0 + 16 + li] = acc1[r];
        }
    }
}

// Trailing update A22 -= L21 L21^T on the lower triangle; 32x32 block per workgroup, one
// 16x16 fp64 MFMA tile per wave.
__global__ __launch_bounds__(256) void syrk_trailing_kernel(double* __restrict__ Kall, int N, int k0, int nb, int cend) {
    const int ti = blockIdx.y, tj = blockIdx.x;              // cend: columns >= cend are left to the outer (rank-128) update
    if (tj > ti) return;
    const int a = blockIdx.z;
    double* K = Kall + (size_t)a * N * N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = k0 + nb;
    const int i0 = r0 + ti * 32 + (wave >> 1) * 16;
    const int j0 = r0 + tj * 32 + (wave & 1) * 16;
    if (j0 > i0 + 15) return;
    const int li = lane & 15, lk = lane >> 4;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    const int ri = i0 + li, rj = j0 + li;
    const double* Ar = K + (size_t)(ri < N ? ri : N - 1) * N + k0;
    const double* Br = K + (size_t)(rj < N ? rj : N - 1) * N + k0;
    double cold[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {                            // the old values travel while the products are formed
        const int row = i0 + lk + 4 * r, col = j0 + li;
        cold[r] = (row < N && col <= row && col < cend) ? K[(size_t)row * N + col] : 0.0;
    }
    mfma_kloop<8>(acc, 0, nb, lk, [&](int k) { return ri < N ? Ar[k] : 0.0; }, [&](int k) { return rj < N ? Br[k] : 0.0; });
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = i0 + lk + 4 * r, col = j0 + li;
        if (row < N && col <= row && col < cend) K[(size_t)row * N + col] = cold[r] - acc[r];
    }
}

// Trailing update of panel k FUSED with the factorisation of panel k + 1's diagonal block (32-wide path, round 5 staging): the
// workgroup that owns trailing tile (0, 0) -- rows / columns [k0 + nb, k0 + nb + 32): exactly the next diagonal block -- keeps its
// updated block in LDS instead of writing it back and runs potrf_diag_fast_kernel's register pivot loop on it (all 1024 threads; in
// the other workgroups wavefronts 4 .. 15 leave at once).  One launch and one dependent launch gap less per panel, and the
// factorisation of the next block no longer waits for the slowest trailing tile.
__global__ __launch_bounds__(NB * NB) void syrk_trailing_potrf_kernel(double* __restrict__ Kall, double* __restrict__ Yall,
                                                                     int N, int k0, int nb, int* __restrict__ info) {
    __shared__ double blk[NB][NB + 1];
    __shared__ double colb[2][2 * NB];
    __shared__ double sinv[NB];
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (tj > ti) return;
    const int a = blockIdx.z;
    double* K = Kall + (size_t)a * N * N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool special = (ti == 0 && tj == 0);
    if (!special && wave >= 4) return;
    const int r0 = k0 + nb;
    if (wave < 4) {
        const int i0 = r0 + ti * 32 + (wave >> 1) * 16;
        const int j0 = r0 + tj * 32 + (wave & 1) * 16;
        if (j0 <= i0 + 15) {
            const int li = lane & 15, lk = lane >> 4;
            d4 acc = {0.0, 0.0, 0.0, 0.0};
            const int ri = i0 + li, rj = j0 + li;
            const double* Ar = K + (size_t)(ri < N ? ri : N - 1) * N + k0;
            const double* Br = K + (size_t)(rj < N ? rj : N - 1) * N + k0;
            double cold[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {                            // the old values travel while the products are formed
                const int row = i0 + lk + 4 * r, col = j0 + li;
                cold[r] = (row < N && col <= row) ? K[(size_t)row * N + col] : 0.0;
            }
            mfma_kloop<8>(acc, 0, nb, lk, [&](int k) { return ri < N ? Ar[k] : 0.0; }, [&](int k) { return rj < N ? Br[k] : 0.0; });
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + lk + 4 * r, col = j0 + li;
                if (row < N && col <= row) {
                    if (special) blk[row - r0][col - r0] = cold[r] - acc[r];
                    else K[(size_t)row * N + col] = cold[r] - acc[r];
                }
            }
        }
    }
    if (!special) return;
    __syncthreads();
    // ---- the next panel's diagonal block: potrf_diag_fast_kernel's pivot loop on the block in LDS -----------------------------
    double* Y = Yall + (size_t)a * N * N;
    const int nb2 = (N - r0 < NB) ? (N - r0) : NB;
    const int c = tid >> 5, r = tid & 31;                    // column-major over the wavefronts: wave w <-> columns 2w, 2w + 1
    double a0 = (r < nb2 && c < nb2 && c <= r) ? blk[r][c] : 0.0;
    double a1 = (r == c) ? 1.0 : 0.0;
    if (c == 0) { colb[0][r] = a0; colb[0][NB + r] = a1; }
    if (tid == 0) {
        if (!(a0 > 0.0) && info[a] == 0) info[a] = r0 + 1;
        sinv[0] = inv_sqrt_pos_p(a0);
    }
    if (tid >= nb2 && tid < NB) sinv[tid] = 0.0;
    __syncthreads();
    for (int k = 0; k + 1 < nb2; ++k) {
        if (2 * wave + 1 > k) {                              // wave-uniform: both columns of a finished wave are final
            const double* cb = colb[k & 1];
            double* cn = colb[(k + 1) & 1];
            const double inv = sinv[k];
            if (c > k && c < nb2) {
                const double lc = cb[c] * inv;
                a0 = fma(-(cb[r] * inv), lc, a0);
                a1 = fma(-(cb[NB + r] * inv), lc, a1);
                if (c == k + 1) {
                    cn[r] = a0;
                    cn[NB + r] = a1;
                    if (r == k + 1) {
                        if (!(a0 > 0.0) && info[a] == 0) info[a] = r0 + k + 2;
                        sinv[k + 1] = inv_sqrt_pos_p(a0);
                    }
                }
            }
        }
        __syncthreads();
    }
    const double sc = sinv[c];                               // 0 for c >= nb2
    if (r < nb2 && c <= r) K[(size_t)(r0 + r) * N + r0 + c] = a0 * sc;
    if (r < nb2 && c < nb2) Y[(size_t)(r0 + c) * N + r0 + r] = (r <= c) ? a1 * sc : 0.0;       // row r of the identity -> column r of Y11
}

// Row block k of Y = L^-1:  Y[k, c] = -Ykk * (sum_p L[k, p] Y[p, c]) for column tiles c < k0.
// blocks > 0: one launch for ALL outer blocks of `blocks` rows (blockIdx.z = outer block K): row block k0 = K + koff of
// each, columns [K, k0) only -- the diagonal blocks Y_KK of the recursive-doubling inverse below.
__global__ __launch_bounds__(256) void trinv_row_kernel(const double* __restrict__ Kall, double* __restrict__ Yall,
                                                        int N, int k0, int nb, int cbeg, int blocks, int koff) {
    __shared__ double w[32][33];
    __shared__ double ykk[NB][NB + 1];
    if (blocks > 0) {
        cbeg = blockIdx.z * blocks;
        k0 = cbeg + koff;
        nb = (N - k0 < NB) ? (N - k0) : NB;
        if (nb <= 0) return;
    }
    const int a = blockIdx.y;
    const double* L = Kall + (size_t)a * N * N;
    double* Y = Yall + (size_t)a * N * N;
    const int c0 = cbeg + blockIdx.x * 32;                   // cbeg > 0: only the columns of the current 128-row outer block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wi = (wave >> 1) * 16, wj = (wave & 1) * 16;
    const int li = lane & 15, lk = lane >> 4;
    {
        const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;          // four loads in flight (see trsm_panel_mfma_kernel)
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + 8 * u;
            const int rr = r < nb ? r : nb - 1, cc = c <= rr ? c : rr;
            v[u] = Y[(size_t)(k0 + rr) * N + (k0 + cc)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + 8 * u;
            ykk[r][c] = (r < nb && c <= r) ? v[u] : 0.0;
        }
    }
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    const bool rowin = (wi + li < nb);
    const int rowA = k0 + (rowin ? wi + li : 0);              // row of L in block k
    const bool colin = (c0 + wj + li < k0);
    const int colB = colin ? c0 + wj + li : 0;                // column of Y
    const double* Ar = L + (size_t)rowA * N;
    const double* Bc = Y + colB;
    mfma_kloop<8>(acc, c0, k0, lk, [&](int pk) { return rowin ? Ar[pk] : 0.0; }, [&](int pk) { return colin ? Bc[(size_t)pk * N] : 0.0; });
#pragma unroll
    for (int r = 0; r < 4; ++r) w[wi + lk + 4 * r][wj + li] = acc[r];
    __syncthreads();
    for (int idx = threadIdx.x; idx < 32 * 32; idx += 256) {
        const int r = idx / 32, c = idx % 32;
        if (r < nb && c0 + c < k0) {
            double s = 0.0;
            for (int m = 0; m <= r; ++m) s = fma(ykk[r][m], w[m][c], s);
            Y[(size_t)(k0 + r) * N + c0 + c] = -s;
        }
    }
}

// Y = L^-1 in ONE launch (32-wide path, round 5 staging): a workgroup per 32-column block c of Y walks the row blocks k = c + 1 ..
// itself -- Y[k, c] = -Y_kk (sum_{p = c .. k-1} L[k, p] Y[p, c]) -- with its block column of Y resident in LDS (rows c0 .. N: at most 544
// rows of 32 doubles), so the 15 dependent launches of trinv_row_kernel (and the events that put them on a side stream) become
// one; the longest chain (c = 0) is 3840 matrix instructions on one CU.  Y_kk is what potrf left in the diagonal blocks.
__global__ __launch_bounds__(256) void trinv_colblock_kernel(const double* __restrict__ Kall, double* __restrict__ Yall, int N) {
    extern __shared__ __attribute__((aligned(16))) double sm_inv[];
    double (*w)[33] = reinterpret_cast<double (*)[33]>(sm_inv);                 // 32 x 33
    double (*ykk)[NB + 1] = reinterpret_cast<double (*)[NB + 1]>(sm_inv + 32 * 33);   // 32 x 33
    double* ycol = sm_inv + 2 * 32 * 33;                                         // (N - c0) x 32
    const int a = blockIdx.y;
    const double* L = Kall + (size_t)a * N * N;
    double* Y = Yall + (size_t)a * N * N;
    const int c0 = blockIdx.x * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = (wave >> 1) * 16, wj = (wave & 1) * 16;
    const int li = lane & 15, lk = lane >> 4;
    const int nbc = (N - c0 < NB) ? (N - c0) : NB;
    // block row c: the diagonal block itself
    for (int idx = tid; idx < 32 * 32; idx += 256) {
        const int r = idx >> 5, c = idx & 31;
        ycol[r * 32 + c] = (r < nbc && c < nbc && c <= r) ? Y[(size_t)(c0 + r) * N + c0 + c] : 0.0;
    }
    __syncthreads();
    for (int k0 = c0 + 32; k0 < N; k0 += 32) {
        const int nb = (N - k0 < NB) ? (N - k0) : NB;
        {
            const int c = tid & 31, r0 = tid >> 5;          // four loads in flight (see trsm_panel_mfma_kernel)
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = r0 + 8 * u;
                const int rr = r < nb ? r : nb - 1, cc = c <= rr ? c : rr;
                v[u] = Y[(size_t)(k0 + rr) * N + (k0 + cc)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = r0 + 8 * u;
                ykk[r][c] = (r < nb && c <= r) ? v[u] : 0.0;
            }
        }
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        const bool rowin = (wi + li < nb);
        const double* Ar = L + (size_t)(k0 + (rowin ? wi + li : 0)) * N;
        const double* Bc = ycol + wj + li - (size_t)c0 * 32;                    // Bc[p * 32] = Y[p, c0 + wj + li] for c0 <= p < k0
        mfma_kloop<8>(acc, c0, k0, lk, [&](int pk) { return rowin ? Ar[pk] : 0.0; }, [&](int pk) { return Bc[(size_t)pk * 32]; });
#pragma unroll
        for (int r = 0; r < 4; ++r) w[wi + lk + 4 * r][wj + li] = acc[r];
        __syncthreads();
        for (int idx = tid; idx < 32 * 32; idx += 256) {
            const int r = idx >> 5, c = idx & 31;
            double sv = 0.0;
            if (r < nb && c < nbc) {
                for (int m = 0; m <= r; ++m) sv = fma(ykk[r][m], w[m][c], sv);
                Y[(size_t)(k0 + r) * N + c0 + c] = -sv;
            }
            ycol[(size_t)(k0 - c0 + r) * 32 + c] = (r < nb && c < nbc) ? -sv : 0.0;
        }
        __syncthreads();
    }
}

// Row blocks [kb, ke) of Y = L^-1 in one launch (32-wide path, round 5 staging): the workgroup of column tile c walks the row blocks
// k = max(kb, c + 1) .. ke - 1 in turn; rows of its tile computed by EARLIER launches come from global memory, the rows of this
// launch stay in LDS.  With four row blocks per launch the inverse's side-stream chain is 4 launches and 8 event operations at N = 500
// instead of 15 and 30 -- the chain of the 32-wide path is paced by the host's enqueue rate.
constexpr int kInvBatchMax = 4;                        // 32 KB of rows + 17 KB of scratch: static LDS
__global__ __launch_bounds__(256) void trinv_rows_batch_kernel(const double* __restrict__ Kall, double* __restrict__ Yall, int N, int kb, int ke) {
    __shared__ double w[32][33];
    __shared__ double ykk[NB][NB + 1];
    __shared__ double yl[kInvBatchMax * 32 * 32];            // rows [kb0, ke0) of this column tile
    const int a = blockIdx.y;
    const double* L = Kall + (size_t)a * N * N;
    double* Y = Yall + (size_t)a * N * N;
    const int c = blockIdx.x, c0 = c * 32, kb0 = kb * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = (wave >> 1) * 16, wj = (wave & 1) * 16;
    const int li = lane & 15, lk = lane >> 4;
    const int nbc = (N - c0 < NB) ? (N - c0) : NB;
    if (c >= kb) {                                           // the tile's diagonal block belongs to this launch's rows: Y_cc from potrf
        for (int idx = tid; idx < 32 * 32; idx += 256) {
            const int r = idx >> 5, cc = idx & 31;
            yl[(size_t)(c0 - kb0 + r) * 32 + cc] = (r < nbc && cc < nbc && cc <= r) ? Y[(size_t)(c0 + r) * N + c0 + cc] : 0.0;
        }
    }
    __syncthreads();
    for (int k = (kb > c + 1 ? kb : c + 1); k < ke; ++k) {
        const int k0 = k * 32;
        const int nb = (N - k0 < NB) ? (N - k0) : NB;
        {
            const int cc = tid & 31, r0 = tid >> 5;          // four loads in flight (see trsm_panel_mfma_kernel)
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = r0 + 8 * u;
                const int rr = r < nb ? r : nb - 1, c2 = cc <= rr ? cc : rr;
                v[u] = Y[(size_t)(k0 + rr) * N + (k0 + c2)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = r0 + 8 * u;
                ykk[r][cc] = (r < nb && cc <= r) ? v[u] : 0.0;
            }
        }
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        const bool rowin = (wi + li < nb);
        const double* Ar = L + (size_t)(k0 + (rowin ? wi + li : 0)) * N;
        const int colg = c0 + wj + li;
        const bool colin = colg < N;
        const double* Bg = Y + (colin ? colg : 0);
        const int split = c0 > kb0 ? c0 : kb0;               // [c0, split): rows of earlier launches; [split, k0): rows of this one (LDS)
        if (c0 < kb0)
            mfma_kloop<8>(acc, c0, kb0 < k0 ? kb0 : k0, lk, [&](int pk) { return rowin ? Ar[pk] : 0.0; }, [&](int pk) { return colin ? Bg[(size_t)pk * N] : 0.0; });
        const double* Bl = yl + wj + li - (size_t)kb0 * 32;
        mfma_kloop<8>(acc, split, k0, lk, [&](int pk) { return rowin ? Ar[pk] : 0.0; }, [&](int pk) { return Bl[(size_t)pk * 32]; });
#pragma unroll
        for (int r = 0; r < 4; ++r) w[wi + lk + 4 * r][wj + li] = acc[r];
        __syncthreads();
        for (int idx = tid; idx < 32 * 32; idx += 256) {
            const int r = idx >> 5, cc = idx & 31;
            double sv = 0.0;
            if (r < nb && cc < nbc) {
                for (int m = 0; m <= r; ++m) sv = fma(ykk[r][m], w[m][cc], sv);
                Y[(size_t)(k0 + r) * N + c0 + cc] = -sv;
            }
            yl[(size_t)(k0 - kb0 + r) * 32 + cc] = (r < nb && cc < nbc) ? -sv : 0.0;
        }
        __syncthreads();
    }
}

// z = Y y  (wave per row), beta = Y^T z  (thread per column)
// targets (N, D) -> (D, N): the rows of Y are multiplied with ONE column of the targets; read in place, the 64 lanes of a load
// touch 64 cache lines (stride D doubles)
__global__ __launch_bounds__(256) void targets_by_gp_kernel(const double* __restrict__ Ymem, int N, int D, double* __restrict__ yt) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= N * D) return;
    const int a = idx / N, p = idx - a * N;
    yt[idx] = Ymem[(size_t)p * D + a];
}

__global__ __launch_bounds__(256) void zvec_kernel(const double* __restrict__ Yall, const double* __restrict__ yt,
                                                   int N, double* __restrict__ z) {
    const int a = blockIdx.y;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= N) return;
    const double* Y = Yall + (size_t)a * N * N + (size_t)row * N;
    const double* y = yt + (size_t)a * N;
    // four independent partial sums keep four loads per lane in flight (fixed order: reproducible)
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int p = lane;
    for (; p + 192 <= row; p += 256) {
        s0 = fma(Y[p], y[p], s0);
        s1 = fma(Y[p + 64], y[p + 64], s1);
        s2 = fma(Y[p + 128], y[p + 128], s2);
        s3 = fma(Y[p + 192], y[p + 192], s3);
    }
    for (; p <= row; p += 64) s0 = fma(Y[p], y[p], s0);
    double s = (s0 + s1) + (s2 + s3);
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) z[(size_t)a * N + row] = s;
}

// beta = Y^T z: a workgroup owns 64 columns, four wavefronts share the rows (p = c + g, c + g + 4, ... in units of the
// column block's first row), four partial sums each, combined in a fixed order.  (One thread per column walked all N rows
// alone: 20 us at N = 200.)
__global__ __launch_bounds__(256) void beta_kernel(const double* __restrict__ Yall, const double* __restrict__ z,
                                                   int N, double* __restrict__ beta) {
    __shared__ double red[4][64];
    const int a = blockIdx.y;
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 64, i = c0 + cl;
    const double* Y = Yall + (size_t)a * N * N;
    const double* za = z + (size_t)a * N;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (i < N) {
        int p = c0 + rg;                                     // rows above the diagonal hold zeros
        for (; p + 12 < N; p += 16) {
            s0 = fma(Y[(size_t)p * N + i], za[p], s0);
            s1 = fma(Y[(size_t)(p + 4) * N + i], za[p + 4], s1);
            s2 = fma(Y[(size_t)(p + 8) * N + i], za[p + 8], s2);
            s3 = fma(Y[(size_t)(p + 12) * N + i], za[p + 12], s3);
        }
        for (; p < N; p += 4) s0 = fma(Y[(size_t)p * N + i], za[p], s0);
    }
    red[rg][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && i < N) beta[(size_t)a * N + i] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

// Large N: beta in two deterministic stages.  One thread per column leaves 1 wavefront per SIMD, each walking up to N rows
// (0.82 ms at N = 4096, D = 16); here a workgroup sums 256 rows of 64 columns (rows above the diagonal block skipped) and
// a second launch adds the row-chunk partials in order.
__global__ __launch_bounds__(256) void beta_partial_kernel(const double* __restrict__ Yall, const double* __restrict__ z, int N,
                                                           double* __restrict__ partial) {
    __shared__ double red[4][64];
    const int a = blockIdx.z, ch = blockIdx.y, cb = blockIdx.x;
    const int nch = gridDim.y;
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = cb * 64 + cl;
    const int pbeg = ch * 256, pend = (pbeg + 256 < N) ? pbeg + 256 : N;
    if (pend <= cb * 64) {                                    // all rows above the columns' diagonal: Y is zero there
        if (rg == 0 && c < N) partial[((size_t)a * nch + ch) * N + c] = 0.0;
        return;
    }
    const double* Y = Yall + (size_t)a * N * N;
    const double* za = z + (size_t)a * N;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (c < N) {
        int p = pbeg + rg;
        for (; p + 12 < pend; p += 16) {
            s0 = fma(Y[(size_t)p * N + c], za[p], s0);
            s1 = fma(Y[(size_t)(p + 4) * N + c], za[p + 4], s1);
            s2 = fma(Y[(size_t)(p + 8) * N + c], za[p + 8], s2);
            s3 = fma(Y[(size_t)(p + 12) * N + c], za[p + 12], s3);
        }
        for (; p < pend; p += 4) s0 = fma(Y[(size_t)p * N + c], za[p], s0);
    }
    red[rg][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && c < N) partial[((size_t)a * nch + ch) * N + c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

__global__ __launch_bounds__(256) void beta_reduce_kernel(const double* __restrict__ partial, int N, int nch, double* __restrict__ beta) {
    const int a = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    double s = 0.0;
    for (int ch = 0; ch < nch; ++ch) s += partial[((size_t)a * nch + ch) * N + i];
    beta[(size_t)a * N + i] = s;
}

// iK = Y^T Y on lower-triangular 32x32 blocks (mirrored on store), plus T = beta beta^T - iK
// with the diagonal halved.
__global__ __launch_bounds__(256) void syrk_inverse_kernel(const double* __restrict__ Yall, const double* __restrict__ beta,
                                                           int N, double* __restrict__ iKall, double* __restrict__ Tall) {
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (tj > ti) return;
    const int a = blockIdx.z;
    const double* Y = Yall + (size_t)a * N * N;
    double* iK = iKall + (size_t)a * N * N;
    double* T = Tall + (size_t)a * (N + kTPadRows) * N;
    const double* be = beta + (size_t)a * N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = ti * 32 + (wave >> 1) * 16;
    const int j0 = tj * 32 + (wave & 1) * 16;
    const int li = lane & 15, lk = lane >> 4;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    const int ci = (i0 + li < N) ? i0 + li : N - 1, cj = (j0 + li < N) ? j0 + li : N - 1;    // clamped: masked on store
    const int pstart = (i0 > j0 ? i0 : j0) & ~3;
    mfma_kloop<8>(acc, pstart, N, lk, [&](int pk) { return Y[(size_t)pk * N + ci]; }, [&](int pk) { return Y[(size_t)pk * N + cj]; });
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = i0 + lk + 4 * r, col = j0 + li;
        if (row < N && col < N && col <= row) {
            const double v = acc[r];
            double t = be[row] * be[col] - v;
            if (row == col) t *= 0.5;
            iK[(size_t)row * N + col] = v;
            // T keeps only its upper triangle (row index <= column index); the rest stays zero
            T[(size_t)col * N + row] = t;
            if (row != col) iK[(size_t)col * N + row] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Incremental factorisation: the reference refactorises from scratch at every control
// step (gp_mpc_controller.py:117) although the memory grows by at most one point per step and the
// hyper-parameters change only after a training round.  When the new (X, Y) is the cached one plus k
// appended points and the hyper-parameters are unchanged, iK and beta are border-updated in O(k N^2):
//   iK' = [[iK + v v^T / s, -v / s], [-v^T / s, 1 / s]],  v = iK k_new,  s = k(x,x) + noise - k_new^T v.
__global__ void prefix_mismatch_kernel(const double* __restrict__ a, const double* __restrict__ b, size_t n,
                                       int* __restrict__ flag) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && a[i] != b[i]) *flag = 1;
}

__global__ __launch_bounds__(256) void kvec_kernel(const double* __restrict__ X, int n, int E, const double* __restrict__ ils2,
                                                   const double* __restrict__ var, int ldk, double* __restrict__ kv,
                                                   const int* __restrict__ skip) {
    if (*skip) return;                                       // the prefix comparison failed: nothing of the update may be written
    const int a = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int e = 0; e < E; ++e) {
        const double d = X[(size_t)i * E + e] - X[(size_t)n * E + e];
        s = fma(d * d, ils2[a * E + e], s);
    }
    kv[(size_t)a * ldk + i] = var[a] * exp(-0.5 * s);
}

// Appending one point to K = L L^T keeps L and adds the row [l^T, d],  l = L^-1 k,  d^2 = k(x,x) + noise - l.l,
// so  L'^-1 = [[L^-1, 0], [u^T]]  with  u = [-L^-T l / d ; 1 / d]  and  iK' = [[iK, 0], [0, 0]] + u u^T,
// beta' = [beta; 0] + u (u.y').  Working from L^-1 (not from iK) keeps the update as accurate as the
// triangular products: the error is O(eps sqrt(cond K)), not O(eps cond K).

// l = L^-1 k  (wave per row, fixed order)
__global__ __launch_bounds__(256) void border_lvec_kernel(const double* __restrict__ linv, const double* __restrict__ kv, int n,
                                                          int ldk, double* __restrict__ lv, const int* __restrict__ skip) {
    if (*skip) return;
    const int a = blockIdx.y;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    const double* r = linv + ((size_t)a * n + row) * n;
    double s = 0.0;
    for (int j = lane; j <= row; j += 64) s = fma(r[j], kv[(size_t)a * ldk + j], s);
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) lv[(size_t)a * ldk + row] = s;
}

// u (n + 1 entries): 64 columns per workgroup, 16 row slices reduced through LDS in a fixed order
__global__ __launch_bounds__(1024) void border_u_kernel(const double* __restrict__ linv, const double* __restrict__ lv, int n,
                                                        int ldk, const double* __restrict__ var, const double* __restrict__ noise,
                                                        double* __restrict__ uv, int* __restrict__ info, const int* __restrict__ skip) {
    __shared__ double part[16][64];
    __shared__ double ssum[16];
    if (*skip) return;
    const int a = blockIdx.y;
    const int t = threadIdx.x, lane = t & 63, slice = t >> 6;
    const double* l = lv + (size_t)a * ldk;
    double q = 0.0;
    for (int i = t; i < n; i += 1024) q = fma(l[i], l[i], q);
    for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off, 64);
    if (lane == 0) ssum[slice] = q;
    const int j = blockIdx.x * 64 + lane;
    double acc = 0.0;
    if (j < n) {
        const double* c = linv + (size_t)a * n * n + j;
        for (int i = j + slice; i < n; i += 16) acc = fma(l[i], c[(size_t)i * n], acc);
    }
    part[slice][lane] = acc;
    __syncthreads();
    if (slice == 0 && j <= n) {
        double ll = 0.0, dot = 0.0;
        for (int k = 0; k < 16; ++k) { ll += ssum[k]; dot += part[k][lane]; }
        const double d2 = var[a] + noise[a] - ll;
        if (!(d2 > 0.0) && j == 0 && info[a] == 0) info[a] = n + 1;
        const double rd = 1.0 / sqrt(d2);
        uv[(size_t)a * ldk + j] = (j < n) ? -rd * dot : rd;
    }
}

// iK', L'^-1 ((n+1) x (n+1), compact) and beta' from the n x n ones.  The workgroups of the first row band also form
// u . y' (fixed order) for beta' = [beta; 0] + u (u . y') -- it was a launch of its own.
__global__ __launch_bounds__(256) void border_apply_kernel(const double* __restrict__ iK, const double* __restrict__ linv,
                                                           const double* __restrict__ beta, const double* __restrict__ uv,
                                                           const double* __restrict__ Y, int D, int n, int ldk,
                                                           double* __restrict__ iKn, double* __restrict__ linvn,
                                                           double* __restrict__ betan, const int* __restrict__ skip) {
    __shared__ double r[4];
    if (*skip) return;
    const int a = blockIdx.z;
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int n1 = n + 1;
    double dot = 0.0;
    if (blockIdx.y == 0) {                                   // uniform per workgroup
        double q = 0.0;
        for (int p = threadIdx.x; p <= n; p += 256) q = fma(uv[(size_t)a * ldk + p], Y[(size_t)p * D + a], q);
        for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off, 64);
        if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = q;
        __syncthreads();
        dot = (r[0] + r[1]) + (r[2] + r[3]);
    }
    if (i > n || j > n) return;
    const double ui = uv[(size_t)a * ldk + i], uj = uv[(size_t)a * ldk + j];
    const bool old = (i < n && j < n);
    const size_t src = ((size_t)a * n + i) * n + j, dst = ((size_t)a * n1 + i) * n1 + j;
    iKn[dst] = fma(ui, uj, old ? iK[src] : 0.0);                    // exactly symmetric
    linvn[dst] = old ? linv[src] : (i == n ? uj : 0.0);
    if (i == 0) betan[(size_t)a * n1 + j] = fma(uj, dot, j < n ? beta[(size_t)a * n + j] : 0.0);
}

// T from externally supplied iK / beta (gpmpc_set_factors)
__global__ __launch_bounds__(256) void tm_kernel(const double* __restrict__ iK, const double* __restrict__ beta, int N,
                                                 double* __restrict__ T, const int* __restrict__ skip) {
    if (skip && *skip) return;
    const int a = blockIdx.z;
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= N + kTPadRows || j >= N) return;                // the launch covers the zero rows after T_a too (no memset)
    double t = 0.0;
    if (i <= j) {
        t = beta[(size_t)a * N + i] * beta[(size_t)a * N + j] - iK[((size_t)a * N + i) * N + j];
        if (i == j) t *= 0.5;
    }
    T[((size_t)a * (N + kTPadRows) + i) * N + j] = t;        // upper triangle only
}

// ------------------------------------------------------------------------------------------
int grow(Handle* h, Buf& b, size_t need) {
    if (b.p && need <= b.cap) return GPMPC_OK;
    if (b.p) GPMPC_HIP_CHECK(h, hipFree(b.p));
    b.p = nullptr; b.cap = 0;
    GPMPC_HIP_CHECK(h, hipMalloc(&b.p, (need ? need : 1) * sizeof(double)));
    b.cap = need ? need : 1;
    return GPMPC_OK;
}

static inline bool fits(const Buf& b, size_t need) { return b.p && need <= b.cap; }

int ensure_model_buffers(Handle* h, int N, int D, int E, bool need_factor_ws) {
    // allocate with head-room for 64 more points: appended points then reuse the buffers (and their contents)
    const size_t Nc = (size_t)N + 64;
    const bool regrow = !fits(h->iK, (size_t)D * N * N) || !fits(h->Tm, (size_t)D * (N + kTPadRows) * N) ||
                        !fits(h->beta, (size_t)D * N) || !fits(h->Xt, (size_t)E * N) ||
                        (need_factor_ws && (!fits(h->gram, (size_t)D * N * N) || !fits(h->linv, (size_t)D * N * N)));
    const size_t Ns = regrow ? Nc : (size_t)N;
    const size_t NN = (size_t)D * Ns * Ns, DN = (size_t)D * Ns, DE = (size_t)D * E, EN = (size_t)E * Ns;
    int rc;
    if ((rc = grow(h, h->Xt, EN))) return rc;
    if ((rc = grow(h, h->ils2, DE))) return rc;
    if ((rc = grow(h, h->xrange, 2 * (size_t)kMaxE))) return rc;
    if ((rc = grow(h, h->var, kMaxD))) return rc;
    if ((rc = grow(h, h->logvar, kMaxD))) return rc;
    if ((rc = grow(h, h->beta, DN))) return rc;
    if ((rc = grow(h, h->zvec, DN))) return rc;
    if ((rc = grow(h, h->iK, NN))) return rc;
    if ((rc = grow(h, h->Tm, (size_t)D * (Ns + kTPadRows) * Ns))) return rc;    // + zero rows per GP
    if (need_factor_ws) {
        if ((rc = grow(h, h->gram, NN))) return rc;
        if ((rc = grow(h, h->linv, (size_t)D * (Ns + kTPadRows) * Ns))) return rc;   // trades places with Tm in border updates
    }
    if ((rc = grow(h, h->Xc, Ns * E))) return rc;
    if ((rc = grow(h, h->Yc, Ns * D))) return rc;
    if ((rc = grow(h, h->hyp, DE + 2 * (size_t)kMaxD))) return rc;
    if ((rc = grow(h, h->kv, DN))) return rc;
    if ((rc = grow(h, h->vv, DN))) return rc;
    if ((rc = grow(h, h->sc, 2 * (size_t)kMaxD))) return rc;
    if (!h->info) {                                  // info[0 .. kMaxD) | mismatch flag: one read-back covers both
        GPMPC_HIP_CHECK(h, hipMalloc(&h->info, (kMaxD + 1) * sizeof(int)));
        GPMPC_HIP_CHECK(h, hipMemset(h->info, 0, (kMaxD + 1) * sizeof(int)));
        h->mismatch = h->info + kMaxD;
    }
    return GPMPC_OK;
}

static int pack(Handle* h, const double* X, const double* ls, const double* os, int N, int D, int E, hipStream_t s) {
    int n = N * E; if (D * E > n) n = D * E; if (D > n) n = D;
    hipLaunchKernelGGL(pack_inputs_kernel, dim3((n + 255) / 256), dim3(256), 0, s, X, ls, os, N, D, E,
                       h->Xt.p, h->ils2.p, h->var.p, h->logvar.p);
    hipLaunchKernelGGL(xrange_kernel, dim3(E), dim3(256), 0, s, X, N, E, h->xrange.p);
    GPMPC_HIP_CHECK(h, hipGetLastError());
    return GPMPC_OK;
}

static int pack_and_record(Handle* h, const double* X, const double* Y, const double* ls, const double* os, const double* noise,
                           int N, int D, int E, hipStream_t s) {
    int nb = (N * (E > D ? E : D) + 255) / 256;
    if (nb < E) nb = E;
    if (nb > 64) nb = 64;
    if (nb < 1) nb = 1;
    if (nb < E) nb = E;
    hipLaunchKernelGGL(pack_record_kernel, dim3(nb), dim3(256), 0, s, X, Y, ls, os, noise, N, D, E, h->Xt.p, h->ils2.p, h->var.p,
                       h->logvar.p, h->xrange.p, h->Xc.p, h->Yc.p, h->hyp.p);
    GPMPC_HIP_CHECK(h, hipGetLastError());
    h->have_state = true;
    return GPMPC_OK;
}

int run_set_factors(Handle* h, const double* X, const double* iK, const double* beta, const double* ls,
                    const double* os, int N, int D, int E, hipStream_t s) {
    int rc = ensure_model_buffers(h, N, D, E, false);
    if (rc) return rc;
    if ((rc = pack(h, X, ls, os, N, D, E, s))) return rc;
    GPMPC_HIP_CHECK(h, hipMemcpyAsync(h->iK.p, iK, (size_t)D * N * N * sizeof(double), hipMemcpyDeviceToDevice, s));
    GPMPC_HIP_CHECK(h, hipMemcpyAsync(h->beta.p, beta, (size_t)D * N * sizeof(double), hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(tm_kernel, dim3((N + 63) / 64, (N + kTPadRows + 3) / 4, D), dim3(256), 0, s, h->iK.p, h->beta.p, N, h->Tm.p, nullptr);
    GPMPC_HIP_CHECK(h, hipGetLastError());
    h->N = N; h->D = D; h->E = E; h->ready = true;
    h->have_state = false;                      // no (X, Y, hyper-parameters) record for these factors
    return GPMPC_OK;
}

// ------------------------------------------------------------------------------------------
// Exact marginal log-likelihood of the D GPs and its gradient wrt (lengthscales, outputscale, noise): what
// gpytorch's ExactMarginalLogLikelihood + autograd give the reference's training loop (gp_model.py:262-275).
// With Q = beta beta^T - iK (T_a holds its upper triangle, diagonal halved):  d mll / d theta = 1/2 tr(Q dK/dtheta),
//   dK/dl_e = K' (x_ie - x_je)^2 / l_e^3,   dK/dsigma^2 = K' / sigma^2,   dK/dnoise = I      (K' = K without noise).
// Tiles of the upper block triangle as in gram_kernel; K' is recomputed, never stored.  Per-block partial sums
// are written out and added in a fixed order by mll_finish_kernel (bitwise reproducible).
template <int EP>
__global__ __launch_bounds__(256) void mll_tile_kernel(const double* __restrict__ Xt, const double* __restrict__ ils2,
                                                       const double* __restrict__ var, const double* __restrict__ Tm,
                                                       int N, int E, int tpad, double* __restrict__ partial) {
    __shared__ double xi[64][EP + 1];
    __shared__ double red[4][EP + 1];
    const int a = blockIdx.z;
    const int ti = blockIdx.y, tj = blockIdx.x;
    const int nt = gridDim.x;
    double* out = partial + ((size_t)a * nt * nt + (size_t)ti * nt + tj) * (EP + 1);
    if (tj < ti) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = ti * 64, j0 = tj * 64;
    for (int idx = threadIdx.x; idx < 64 * EP; idx += 256) {
        const int r = idx / EP, e = idx - r * EP;
        const int i = i0 + r;
        xi[r][e] = (e < E && i < N) ? Xt[(size_t)e * N + i] * sqrt(ils2[a * E + e]) : 0.0;
    }
    const int j = j0 + lane;
    double xj[EP], acc[EP + 1];
#pragma unroll
    for (int e = 0; e < EP; ++e) {
        xj[e] = (e < E && j < N) ? Xt[(size_t)e * N + j] * sqrt(ils2[a * E + e]) : 0.0;
        acc[e] = 0.0;
    }
    acc[EP] = 0.0;
    __syncthreads();
    const double va = var[a];
    const double* Ta = Tm + (size_t)a * (N + tpad) * N;
    for (int rr = 0; rr < 16; ++rr) {
        const int r = wave * 16 + rr;
        const int i = i0 + r;
        double d2[EP];
        double s = 0.0;
#pragma unroll
        for (int e = 0; e < EP; ++e) { const double d = xi[r][e] - xj[e]; d2[e] = d * d; s += d2[e]; }
        const double q = (i < N && j < N) ? 2.0 * Ta[(size_t)i * N + j] : 0.0;        // zero below the diagonal
        const double wq = q * va * exp(-0.5 * s);
#pragma unroll
        for (int e = 0; e < EP; ++e) acc[e] = fma(wq, d2[e], acc[e]);
        acc[EP] += wq;
    }
#pragma unroll
    for (int k = 0; k <= EP; ++k) {
        double v = acc[k];
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x <= EP) out[threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// out[a] = [nll | d nll / d l_e (E) | d nll / d outputscale | d nll / d noise],  nll = -mll / N
__global__ __launch_bounds__(256) void mll_finish_kernel(const double* __restrict__ partial, int nt, int EP,
                                                         const double* __restrict__ Y, const double* __restrict__ beta,
                                                         const double* __restrict__ L, const double* __restrict__ Tm,
                                                         const double* __restrict__ ils2, const double* __restrict__ var,
                                                         int N, int D, int E, int tpad, double* __restrict__ out) {
    __shared__ double red[3][4];
    __shared__ double sums[32];
    const int a = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double yb = 0.0, ld = 0.0, tq = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) {
        yb = fma(Y[(size_t)i * D + a], beta[(size_t)a * N + i], yb);
        ld += log(L[((size_t)a * N + i) * N + i]);
        tq += 2.0 * Tm[((size_t)a * (N + tpad) + i) * N + i];
    }
    for (int off = 32; off >= 1; off >>= 1) {
        yb += __shfl_xor(yb, off, 64); ld += __shfl_xor(ld, off, 64); tq += __shfl_xor(tq, off, 64);
    }
    if (lane == 0) { red[0][wave] = yb; red[1][wave] = ld; red[2][wave] = tq; }
    if (threadIdx.x <= EP) {                    // block partials of the upper block triangle, fixed order
        double v = 0.0;
        for (int ti = 0; ti < nt; ++ti)
            for (int tj = ti; tj < nt; ++tj) v += partial[((size_t)a * nt * nt + (size_t)ti * nt + tj) * (EP + 1) + threadIdx.x];
        sums[threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double ybs = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        const double lds = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        const double tqs = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
        double* o = out + (size_t)a * (E + 3);
        const double invn = 1.0 / (double)N;
        o[0] = (0.5 * ybs + lds + 0.5 * N * 1.8378770664093453) * invn;           // log(2 pi)
        for (int e = 0; e < E; ++e) o[1 + e] = -0.5 * invn * sums[e] * sqrt(ils2[a * E + e]);     // S_e / l_e
        o[1 + E] = -0.5 * invn * sums[EP] / var[a];
        o[2 + E] = -0.5 * invn * tqs;
    }
}

// remember what the cached factors were computed from
// pivot info read back from the device -> error state (info is cleared again: zero when idle)
static int check_info_host(Handle* h, const int* info, int D, hipStream_t s) {
    for (int a = 0; a < D; ++a) {
        if (info[a] != 0) {
            char buf[160];
            snprintf(buf, sizeof buf, "cholesky: GP %d: leading minor of order %d of K + noise*I is not positive-definite",
                     a, info[a]);
            h->err = buf;
            h->ready = false;
            h->have_state = false;
            (void)hipMemsetAsync(h->info, 0, kMaxD * sizeof(int), s);          // zero when idle
            return GPMPC_ERR_NOT_PD;
        }
    }
    return GPMPC_OK;
}

static int check_info(Handle* h, int D, hipStream_t s) {
    int info[kMaxD];
    GPMPC_HIP_CHECK(h, hipMemcpyAsync(info, h->info, kMaxD * sizeof(int), hipMemcpyDeviceToHost, s));
    GPMPC_HIP_CHECK(h, hipStreamSynchronize(s));
    return check_info_host(h, info, D, s);
}

// Border-update / reuse of the cached factors; returns 1 if it handled the call, 0 to fall through to the
// full factorisation, < 0 on error.
static int try_incremental(Handle* h, const double* X, const double* Y, const double* ls, const double* os,
                           const double* noise, int N, int D, int E, hipStream_t s) {
    if (!h->opt_incremental || !h->ready || !h->have_state || D != h->D || E != h->E) return 0;
    const int n0 = h->N, k = N - n0;
    if (k < 0 || k > 8 || h->inc_updates + k > h->opt_refresh_every) return 0;
    const size_t NN = (size_t)D * N * N, DN = (size_t)D * N;
    const size_t TN = (size_t)D * (N + kTPadRows) * N;
    if (!fits(h->iK, NN) || !fits(h->gram, NN) || !fits(h->linv, TN) || !fits(h->beta, DN) || !fits(h->zvec, DN) ||
        !fits(h->Tm, TN) || !fits(h->Xt, (size_t)E * N) || !fits(h->Xc, (size_t)N * E) ||
        !fits(h->Yc, (size_t)N * D) || !fits(h->kv, DN) || !fits(h->vv, DN))
        return 0;
    // is the cached (X, Y, hyper-parameters) a prefix of the new one?  (flag is zero when idle)
    {
        const size_t nX = (size_t)n0 * E, nY = (size_t)n0 * D;
        int nb = (int)((nX + 255) / 256);
        if (nb > 64) nb = 64;
        if (nb < 1) nb = 1;
        hipLaunchKernelGGL(prefix_mismatch_all_kernel, dim3(nb), dim3(256), 0, s, X, h->Xc.p, nX, Y, h->Yc.p, nY, ls, os, noise,
                           h->hyp.p, D, E, h->mismatch);
    }
    if (k == 0) {                                                  // nothing appended: cache hit if the comparison says so
        int flag = 1;
        GPMPC_HIP_CHECK(h, hipMemcpyAsync(&flag, h->mismatch, sizeof(int), hipMemcpyDeviceToHost, s));
        GPMPC_HIP_CHECK(h, hipStreamSynchronize(s));
        if (flag) { GPMPC_HIP_CHECK(h, hipMemsetAsync(h->mismatch, 0, sizeof(int), s)); return 0; }
        h->last_prepare_mode = 2;
        return 1;
    }
    // The border update is enqueued behind the comparison without waiting for its verdict: every kernel of it returns at once
    // when the flag is set, so a failed comparison leaves the cached factors untouched.  ONE read-back (flag + pivot info)
    // at the end decides (was: a synchronisation for the flag, then one for the pivots).
    const Buf b_iK = h->iK, b_gram = h->gram, b_linv = h->linv, b_Tm = h->Tm, b_beta = h->beta, b_zvec = h->zvec;
    for (int n = n0; n < N; ++n) {                                  // (info is zero when idle: check_info clears it after a failure)
        hipLaunchKernelGGL(kvec_kernel, dim3((n + 255) / 256, D), dim3(256), 0, s, X, n, E, h->ils2.p, h->var.p, N, h->kv.p, h->mismatch);
        hipLaunchKernelGGL(border_lvec_kernel, dim3((n + 3) / 4, D), dim3(256), 0, s, h->linv.p, h->kv.p, n, N, h->vv.p, h->mismatch);
        hipLaunchKernelGGL(border_u_kernel, dim3((n + 64) / 64, D), dim3(1024), 0, s, h->linv.p, h->vv.p, n, N, h->var.p, noise,
                           h->kv.p, h->info, h->mismatch);                           // u overwrites k
        hipLaunchKernelGGL(border_apply_kernel, dim3((n + 64) / 64, (n + 4) / 4, D), dim3(256), 0, s, h->iK.p, h->linv.p,
                           h->beta.p, h->kv.p, Y, D, n, N, h->gram.p, h->Tm.p, h->zvec.p, h->mismatch);
        Buf t = h->iK; h->iK = h->gram; h->gram = t;
        t = h->linv; h->linv = h->Tm; h->Tm = t;
        t = h->beta; h->beta = h->zvec; h->zvec = t;
    }
    hipLaunchKernelGGL(tm_kernel, dim3((N + 63) / 64, (N + kTPadRows + 3) / 4, D), dim3(256), 0, s, h->iK.p, h->beta.p, N, h->Tm.p,
                       h->mismatch);
    GPMPC_HIP_CHECK(h, hipGetLastError());
    int verdict[kMaxD + 1];
    GPMPC_HIP_CHECK(h, hipMemcpyAsync(verdict, h->info, (kMaxD + 1) * sizeof(int), hipMemcpyDeviceToHost, s));
    GPMPC_HIP_CHECK(h, hipStreamSynchronize(s));
    if (verdict[kMaxD]) {                                          // not a prefix: nothing was written; back to the full path
        h->iK = b_iK; h->gram = b_gram; h->linv = b_linv; h->Tm = b_Tm; h->beta = b_beta; h->zvec = b_zvec;
        GPMPC_HIP_CHECK(h, hipMemsetAsync(h->mismatch, 0, sizeof(int), s));
        return 0;
    }
    int rc = check_info_host(h, verdict, D, s);
    if (rc) return rc;
    if ((rc = pack_and_record(h, X, Y, ls, os, noise, N, D, E, s))) return rc;
    h->N = N;
    h->inc_updates += k;
    h->last_prepare_mode = 1;
    return 1;
}

int run_prepare(Handle* h, const double* X, const double* Y, const double* ls, const double* os,
                const double* noise, int N, int D, int E, hipStream_t s) {
    h->last_prepare_mode = 0;
    int rc = try_incremental(h, X, Y, ls, os, noise, N, D, E, s);
    if (rc != 0) return rc < 0 ? rc : GPMPC_OK;
    rc = ensure_model_buffers(h, N, D, E, true);
    if (rc) return rc;
    h->ready = false;
    h->have_state = false;
    // small memories: one launch, one workgroup per GP (prepare_small.hip) -- everything, or (rc == 2) the factorisation only
    rc = run_prepare_small(h, X, Y, ls, os, noise, N, D, E, s);
    if (rc < 0) return rc;
    if (rc == 1) {
        if ((rc = check_info(h, D, s))) return rc;
        h->have_state = true;                    // the kernel recorded (X, Y, hyper-parameters)
        h->inc_updates = 0;
        h->N = N; h->D = D; h->E = E; h->ready = true;
        return GPMPC_OK;
    }
    const bool factored = (rc == 2);
    if (!factored) {
        if ((rc = pack_and_record(h, X, Y, ls, os, noise, N, D, E, s))) return rc;
        h->have_state = false;                   // valid only once the factorisation has succeeded
        GPMPC_HIP_CHECK(h, hipMemsetAsync(h->linv.p, 0, (size_t)D * N * N * sizeof(double), s));
    }
    if (!factored) GPMPC_HIP_CHECK(h, hipMemsetAsync(h->Tm.p, 0, (size_t)D * (N + kTPadRows) * N * sizeof(double), s));
    if (!factored) {
        const dim3 grid((N + 63) / 64, (N + 63) / 64, D);
        const int lower = (N >= h->opt_outer_min_n && h->opt_outer_block != 0) ? 1 : 0;
        const int nt_l = (N + 63) / 64;
        // all GPs per tile, squared differences shared (gram_lower_kernel), once its one-workgroup-per-tile grid fills the chip
        // (measured, profiles/r06_gram_ab.txt: N = 4096, D = 16: 985 -> 264 us = 4.1 TB/s of stores; N = 1000, D = 4 (136 tiles):
        // 15.0 -> 17.4 us, hence the tile-count rule); "gram_shared" 0: the per-GP kernel always, 2: the shared one whenever lower (A/B)
        if (lower && (h->opt_gram_shared == 2 || (h->opt_gram_shared == 1 && nt_l * (nt_l + 1) / 2 >= 2 * h->num_cu))) {
            const int nt = nt_l;
            const dim3 gl(nt * (nt + 1) / 2);
            auto go = [&](auto kern) { hipLaunchKernelGGL(kern, gl, dim3(256), 0, s, h->Xt.p, h->ils2.p, h->var.p, noise, N, E, D, h->gram.p); };
            if (E <= 4) go(gram_lower_kernel<4>);
            else if (E <= 8) go(gram_lower_kernel<8>);
            else if (E <= 12) go(gram_lower_kernel<12>);
            else if (E <= 16) go(gram_lower_kernel<16>);
            else if (E <= 20) go(gram_lower_kernel<20>);
            else go(gram_lower_kernel<24>);
        }
        else if (E <= 4) hipLaunchKernelGGL(gram_kernel<4>, grid, dim3(256), 0, s, h->Xt.p, h->ils2.p, h->var.p, noise, N, E, h->gram.p, lower);
        else if (E <= 8) hipLaunchKernelGGL(gram_kernel<8>, grid, dim3(256), 0, s, h->Xt.p, h->ils2.p, h->var.p, noise, N, E, h->gram.p, lower);
        else if (E <= 16) hipLaunchKernelGGL(gram_kernel<16>, grid, dim3(256), 0, s, h->Xt.p, h->ils2.p, h->var.p, noise, N, E, h->gram.p, lower);
        else hipLaunchKernelGGL(gram_kernel<24>, grid, dim3(256), 0, s, h->Xt.p, h->ils2.p, h->var.p, noise, N, E, h->gram.p, lower);
        GPMPC_HIP_CHECK(h, hipGetLastError());
    }
    // Outer blocking (large N): the rank-32 trailing update touches the whole trailing matrix per panel -- 32 multiply-adds
    // per 16 bytes read and written, HBM-bound at N = 4096.  With outer panels of 128 columns the 32-wide steps only update
    // the strip inside the outer panel and one LDS-tiled rank-128 product per outer panel does the rest.
    const int OW = (N >= h->opt_outer_min_n && h->opt_outer_block != 0) ? 128 : 0;
    // the 128 x 128 tiled kernels address a GP's matrix with 32-bit byte offsets (buffer loads): N^2 * 8 < 4 GiB, N <= 23170;
    // beyond that the 64 x 64 kernels with 64-bit addresses run
    const bool tile128 = h->opt_tile128 != 0 && (size_t)N * N * sizeof(double) < 0xFFFFFFFFull;
    // trailing update after the outer panel that ends at column cend (128 x 128 tiles, binary outer levels)
    auto outer_update = [&](int cend) -> int {
        // Binary outer levels: after m = cend / 128 outer panels, with 2^t the largest power of two dividing m
        // (t <= tmax), the last 2^t panels update the next 2^t tile columns in one product (k = 128 * 2^t) -- or,
        // at t = tmax, everything to the right.  Every element of the trailing matrix is then read and written
        // once per 128 * 2^tmax columns instead of once per 128 (the update is bound by that traffic).
        const int nto = (N - cend + T2 - 1) / T2;
        const int tmax = h->opt_outer2 < 0 ? 0 : (h->opt_outer2 > 4 ? 4 : h->opt_outer2);
        const int m = cend / OW;
        int t = 0;
        while (t < tmax && (m & ((2 << t) - 1)) == 0) ++t;
        int nct = (t == tmax) ? nto : (1 << t);
        if (nct > nto) nct = nto;
        const int ntile = nct * nto - nct * (nct - 1) / 2;
        const int wk = OW << t;
        hipLaunchKernelGGL(syrk_outer_t128_kernel, dim3(ntile * D), dim3(512), 0, s, h->gram.p, N, D, ntile,
                           cend - wk, wk, nto);
        GPMPC_HIP_CHECK(h, hipGetLastError());
        return GPMPC_OK;
    };
    // 32-wide panel path (256 < N < outer_min_n): row block k of Y = L^-1 needs rows <= k of L and the earlier rows of Y only --
    // not the panel solve and trailing update of step k -- so the inverse's chain of launches runs on a side stream BESIDE the
    // factorisation's (one event per step; every kernel here fills a few CUs).  N = 500: the factorisation chain is 16 x
    // (8.9 + 4.7 + 5.0) us, the inverse chain 15 x 14.3 us (profiles/r04_c3_kernel_trace_stats.txt); in sequence 0.59 ms.
    // ... or, up to 544 points (17 row blocks of the block column in LDS), the whole inverse as ONE launch after the factorisation
    const bool inv_cols = !OW && !factored && h->opt_outer_block != 0 && h->opt_prepare_invcols != 0 && N > NB &&
                          N <= (h->opt_prepare_invcols == 2 ? 544 : 352);       // measured: 0.286 -> 0.254 ms at N = 257, 0.315 -> 0.297 at 300, 0.404 vs 0.420 at 400, slower from 500 on
    const bool overlap_inv = !OW && !factored && !inv_cols && h->opt_prepare_overlap != 0 && N > NB;
    if (overlap_inv && !h->side_stream) {
        GPMPC_HIP_CHECK(h, hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
        GPMPC_HIP_CHECK(h, hipEventCreateWithFlags(&h->ev_params, hipEventDisableTiming));
        GPMPC_HIP_CHECK(h, hipEventCreateWithFlags(&h->ev_points, hipEventDisableTiming));
    }
    int inv_rows_done = 1;                                // row blocks [1, inv_rows_done) of L^-1 are launched (block 0 is its diagonal block)
    for (int k0 = 0; k0 < N; k0 += NB) {
        const int nb = (N - k0 < NB) ? (N - k0) : NB;
        const bool blk128 = OW && tile128 && h->opt_block128 != 0;
        if (!factored && blk128) {
            // whole outer panel: block factorisation (+ its inverse) and the solve of the rows below, two launches
            if (k0 % OW == 0) {
                const void* kern = reinterpret_cast<const void*>(potrf_block128_kernel);
                if ((rc = allow_full_lds(h, kern))) return rc;
                const size_t lds = (size_t)(T2 * kPS + NB * 33 + 4 * NB + NB) * sizeof(double);
                hipLaunchKernelGGL(potrf_block128_kernel, dim3(D), dim3(1024), lds, s, h->gram.p, h->linv.p, N, k0, h->info);
                const int cend = k0 + OW;
                if (cend < N) {
                    const int nrow = (N - cend + T2 - 1) / T2;
                    hipLaunchKernelGGL(trsm_outer_t128_kernel, dim3(nrow * D), dim3(512), 0, s, h->gram.p, h->linv.p, N, D,
                                       nrow, k0, OW);
                    if ((rc = outer_update(cend))) return rc;
                }
            }
        } else if (!factored) {
            const bool fast = h->opt_outer_block != 0;           // round-2 panel kernels (option "outer_block" = 0: the round-1 ones)
            // inside an outer panel the 32-column panels are factorised left-looking (option "inner_left" = 0: right-looking
            // with a rank-32 update of the rest of the outer panel per step)
            const bool ll = OW && h->opt_inner_left != 0;
            const int left = ll ? k0 % OW : 0;
            // 32-wide path: from the second panel on the diagonal block was factorised by the previous panel's fused trailing update
            const bool fuse_next = fast && !OW && h->opt_prepare_fuse != 0;
            if (fuse_next && k0 > 0) { /* done by syrk_trailing_potrf_kernel of the panel before */ }
            else if (fast) hipLaunchKernelGGL(potrf_diag_fast_kernel, dim3(D), dim3(NB * NB), 0, s, h->gram.p, h->linv.p, N, k0, nb, h->info, left);
            else hipLaunchKernelGGL(potrf_diag_kernel, dim3(D), dim3(NB * NB), 0, s, h->gram.p, h->linv.p, N, k0, nb, h->info);
            if (overlap_inv && k0 > 0) {
                // rows <= k of L and Y_kk are final: row block k of the inverse can start now, beside this step's solve and update --
                // in batches of `prepare_inv_batch` row blocks per launch (1: a launch per row block, the round-4 form)
                const int kp = k0 / NB, last = (N + NB - 1) / NB - 1;
                int batch = h->opt_prepare_inv_batch < 1 ? 1 : (h->opt_prepare_inv_batch > kInvBatchMax ? kInvBatchMax : h->opt_prepare_inv_batch);
                if (!fast) batch = 1;
                if (batch == 1) {
                    GPMPC_HIP_CHECK(h, hipEventRecord(h->ev_params, s));
                    GPMPC_HIP_CHECK(h, hipStreamWaitEvent(h->side_stream, h->ev_params, 0));
                    hipLaunchKernelGGL(trinv_row_kernel, dim3((k0 + 31) / 32, D), dim3(256), 0, h->side_stream, h->gram.p, h->linv.p, N, k0, nb, 0, 0, 0);
                } else if (kp - inv_rows_done + 1 >= batch || kp == last) {
                    GPMPC_HIP_CHECK(h, hipEventRecord(h->ev_params, s));
                    GPMPC_HIP_CHECK(h, hipStreamWaitEvent(h->side_stream, h->ev_params, 0));
                    hipLaunchKernelGGL(trinv_rows_batch_kernel, dim3(kp, D), dim3(256), 0, h->side_stream, h->gram.p, h->linv.p, N, inv_rows_done, kp + 1);
                    inv_rows_done = kp + 1;
                }
            }
            const int M = N - k0 - nb;
            if (M > 0) {
                if (fast && left > 0) hipLaunchKernelGGL(trsm_panel_ll_kernel, dim3((M + 63) / 64, D), dim3(256), 0, s, h->gram.p, h->linv.p, N, k0, nb, left);
                else if (fast) hipLaunchKernelGGL(trsm_panel_mfma_kernel, dim3((M + 63) / 64, D), dim3(256), 0, s, h->gram.p, h->linv.p, N, k0, nb);
                else hipLaunchKernelGGL(trsm_panel_kernel, dim3((M + 255) / 256, D), dim3(256), 0, s, h->gram.p, N, k0, nb);
                int cend = N;
                if (OW) { cend = (k0 / OW + 1) * OW; if (cend > N) cend = N; }
                const int nt = (M + 31) / 32;
                const int ntx = (cend - (k0 + nb) + 31) / 32;
                if (ntx > 0 && !ll) {
                    if (fuse_next) hipLaunchKernelGGL(syrk_trailing_potrf_kernel, dim3(nt, nt, D), dim3(NB * NB), 0, s, h->gram.p, h->linv.p, N, k0, nb, h->info);
                    else hipLaunchKernelGGL(syrk_trailing_kernel, dim3(ntx < nt ? ntx : nt, nt, D), dim3(256), 0, s, h->gram.p, N, k0, nb, cend);
                }
                if (OW && k0 + nb == cend && cend < N) {             // outer panel [cend - OW, cend) complete: rank-OW update of the rest
                    if (tile128) {
                        if ((rc = outer_update(cend))) return rc;
                    } else {
                        const int nto = (N - cend + TS - 1) / TS;
                        hipLaunchKernelGGL(syrk_outer_kernel, dim3(nto, nto, D), dim3(256), 0, s, h->gram.p, N, cend - OW, OW);
                    }
                }
            }
        }
        if (k0 > 0 && !OW && !factored && !overlap_inv && !inv_cols) {
            hipLaunchKernelGGL(trinv_row_kernel, dim3((k0 + 31) / 32, D), dim3(256), 0, s, h->gram.p, h->linv.p, N, k0, nb, 0, 0, 0);
        }
    }
    if (inv_cols) {
        const void* kern = reinterpret_cast<const void*>(trinv_colblock_kernel);
        if ((rc = allow_full_lds(h, kern))) return rc;
        const int nblk = (N + NB - 1) / NB;
        const size_t lds = (size_t)(2 * 32 * 33 + (size_t)nblk * 32 * 32) * sizeof(double);
        hipLaunchKernelGGL(trinv_colblock_kernel, dim3(nblk, D), dim3(256), lds, s, h->gram.p, h->linv.p, N);
    }
    if (overlap_inv) {                                    // join: what follows reads all of Y
        GPMPC_HIP_CHECK(h, hipEventRecord(h->ev_points, h->side_stream));
        GPMPC_HIP_CHECK(h, hipStreamWaitEvent(s, h->ev_points, 0));
    }
    if (OW && tile128) {
        // Y = L^-1 by recursive doubling.  All diagonal 128-blocks Y_KK at once (the 32-row recursion restricted to the columns
        // of the block: 3 launches), then for b = 128, 256, ...: every pair of adjacent b-blocks [[Y11, 0], [Y21, Y22]] gets
        // Y21 = -Y22 (L21 Y11) from two batched tiled products (scratch W = L21 Y11: the iK buffer, written later).  5 levels
        // at N = 4096, every launch >= 256 workgroups, instead of 31 dependent steps whose first ones fill a few CUs.
        const size_t NN = (size_t)N * N;
        const int nblk = (N + OW - 1) / OW;
        for (int koff = NB; koff < OW && h->opt_block128 == 0; koff += NB)      // (potrf_block128_kernel leaves Y_KK behind)
            hipLaunchKernelGGL(trinv_row_kernel, dim3((koff + 31) / 32, D, nblk), dim3(256), 0, s, h->gram.p, h->linv.p, N, 0, 0, 0, OW, koff);
        for (int b = OW; b < N; b *= 2) {
            const int nq = (N - b + 2 * b - 1) / (2 * b);              // pairs q with (2q + 1) b < N
            const int lastrows = N - (2 * (nq - 1) + 1) * b;
            GemmBatch g;
            g.lda = g.ldb = g.ldc = N;
            g.sa = g.sb = g.sc = NN;
            g.qa = g.qb = g.qc = (size_t)2 * b * ((size_t)N + 1);
            g.nq = nq; g.nbatch = nq * D;
            g.M = b; g.M_last = lastrows < b ? lastrows : b;
            g.NC = b; g.Kd = b;
            g.tiles_x = (b + T2 - 1) / T2;
            g.tiles = g.tiles_x * ((b + T2 - 1) / T2);
            const size_t off21 = (size_t)b * N;                        // block (1, 0) of a pair relative to its block (0, 0)
            const size_t off22 = (size_t)b * N + b;
            const dim3 grid(g.tiles * g.nbatch);
            // W = L21 Y11
            g.A = h->gram.p + off21; g.B = h->linv.p; g.C = h->iK.p + off21;
            g.a_rem = NN - off21;
            g.alpha = 1.0; g.kskip = 1; g.ktri = 0; g.kd_is_m = 0;
            hipLaunchKernelGGL(gemm_nn_t128_kernel, grid, dim3(512), 0, s, g);
            // Y21 = -Y22 W
            g.A = h->linv.p + off22; g.B = h->iK.p + off21; g.C = h->linv.p + off21;
            g.a_rem = NN - off22;
            g.alpha = -1.0; g.kskip = 0; g.ktri = 1; g.kd_is_m = 1;
            hipLaunchKernelGGL(gemm_nn_t128_kernel, grid, dim3(512), 0, s, g);
        }
    } else if (OW) {
        // Y = L^-1 by 128-row blocks: inside a block the 32-row recursion (columns of the block only) gives Y_KK; the part left
        // of the block is two tiled products, W = L[K, c:K] Y[c:K, c] (scratch: the iK buffer, written later) and Y[K, c] = -Y_KK W
        const size_t NN = (size_t)N * N;
        for (int K0 = 0; K0 < N; K0 += OW) {
            const int mb = (N - K0 < OW) ? (N - K0) : OW;
            for (int k0 = K0 + NB; k0 < K0 + mb; k0 += NB) {
                const int nb = (N - k0 < NB) ? (N - k0) : NB;
                hipLaunchKernelGGL(trinv_row_kernel, dim3((k0 - K0 + 31) / 32, D), dim3(256), 0, s, h->gram.p, h->linv.p, N, k0, nb, K0, 0, 0);
            }
            if (K0 > 0) {
                const dim3 grid((K0 + TS - 1) / TS, (mb + TS - 1) / TS, D);
                hipLaunchKernelGGL(gemm_nn_tiled_kernel, grid, dim3(256), 0, s, h->gram.p + (size_t)K0 * N, N, NN, h->linv.p, N, NN,
                                   h->iK.p + (size_t)K0 * N, N, NN, mb, K0, K0, 1.0, 1, 0);
                hipLaunchKernelGGL(gemm_nn_tiled_kernel, grid, dim3(256), 0, s, h->linv.p + (size_t)K0 * N + K0, N, NN,
                                   h->iK.p + (size_t)K0 * N, N, NN, h->linv.p + (size_t)K0 * N, N, NN, mb, K0, mb, -1.0, 0, 1);
            }
        }
    }
    GPMPC_HIP_CHECK(h, hipGetLastError());
    hipLaunchKernelGGL(targets_by_gp_kernel, dim3((N * D + 255) / 256), dim3(256), 0, s, Y, N, D, h->vv.p);      // vv: border-update scratch, free here
    hipLaunchKernelGGL(zvec_kernel, dim3((N + 3) / 4, D), dim3(256), 0, s, h->linv.p, h->vv.p, N, h->zvec.p);
    if (N >= h->opt_outer_min_n && tile128) {
        // partials in the iK buffer (written by the product that follows)
        const int nch = (N + 255) / 256;
        hipLaunchKernelGGL(beta_partial_kernel, dim3((N + 63) / 64, nch, D), dim3(256), 0, s, h->linv.p, h->zvec.p, N, h->iK.p);
        hipLaunchKernelGGL(beta_reduce_kernel, dim3((N + 255) / 256, D), dim3(256), 0, s, h->iK.p, N, nch, h->beta.p);
    } else {
        hipLaunchKernelGGL(beta_kernel, dim3((N + 63) / 64, D), dim3(256), 0, s, h->linv.p, h->zvec.p, N, h->beta.p);
    }
    if (N >= h->opt_outer_min_n && h->opt_outer_block != 0 && tile128) {
        const int nt = (N + T2 - 1) / T2, ntile = nt * (nt + 1) / 2;
        hipLaunchKernelGGL(syrk_inverse_t128_kernel, dim3(ntile * D), dim3(512), 0, s, h->linv.p, h->beta.p, N, D, ntile,
                           h->iK.p, h->Tm.p);
    } else if (N >= 512 && h->opt_outer_block != 0) {
        const int nt = (N + TS - 1) / TS;
        hipLaunchKernelGGL(syrk_inverse_tiled_kernel, dim3(nt, nt, D), dim3(256), 0, s, h->linv.p, h->beta.p, N, h->iK.p, h->Tm.p);
    } else {
        const int nt = (N + 31) / 32;
        hipLaunchKernelGGL(syrk_inverse_kernel, dim3(nt, nt, D), dim3(256), 0, s, h->linv.p, h->beta.p, N, h->iK.p, h->Tm.p);
    }
    GPMPC_HIP_CHECK(h, hipGetLastError());
    if ((rc = check_info(h, D, s))) return rc;
    h->have_state = true;                         // (X, Y, hyper-parameters) were recorded by the first launch
    h->inc_updates = 0;
    h->N = N; h->D = D; h->E = E; h->ready = true;
    return GPMPC_OK;
}

int run_mll(Handle* h, const double* X, const double* Y, const double* ls, const double* os, const double* noise,
            int N, int D, int E, double* out_host, hipStream_t s) {
    const int keep = h->opt_incremental;
    h->opt_incremental = 0;                       // always a fresh factorisation: the hyper-parameters are the variables
    int rc = run_prepare(h, X, Y, ls, os, noise, N, D, E, s);
    h->opt_incremental = keep;
    if (rc) return rc;
    const int nt = (N + 63) / 64;
    const int EP = E <= 4 ? 4 : (E <= 8 ? 8 : (E <= 16 ? 16 : 24));
    const size_t npart = (size_t)D * nt * nt * (EP + 1), nout = (size_t)D * (E + 3);
    if ((rc = grow(h, h->mllws, npart + nout))) return rc;
    double* partial = h->mllws.p;
    double* out = partial + npart;
    const dim3 grid(nt, nt, D);
    if (EP == 4) hipLaunchKernelGGL(mll_tile_kernel<4>, grid, dim3(256), 0, s, h->Xt.p, h->ils2.p, h->var.p, h->Tm.p, N, E, kTPadRows, partial);
    else if (EP == 8) hipLaunchKernelGGL(mll_tile_kernel<8>, grid, dim3(256), 0, s, h->Xt.p, h->ils2.p, h->var.p, h->Tm.p, N, E, kTPadRows, partial);
    else if (EP == 16) hipLaunchKernelGGL(mll_tile_kernel<16>, grid, dim3(256), 0, s, h->Xt.p, h->ils2.p, h->var.p, h->Tm.p, N, E, kTPadRows, partial);
    else hipLaunchKernelGGL(mll_tile_kernel<24>, grid, dim3(256), 0, s, h->Xt.p, h->ils2.p, h->var.p, h->Tm.p, N, E, kTPadRows, partial);
    hipLaunchKernelGGL(mll_finish_kernel, dim3(D), dim3(256), 0, s, partial, nt, EP, Y, h->beta.p, h->gram.p, h->Tm.p, h->ils2.p,
                       h->var.p, N, D, E, kTPadRows, out);
    GPMPC_HIP_CHECK(h, hipGetLastError());
    GPMPC_HIP_CHECK(h, hipMemcpyAsync(out_host, out, nout * sizeof(double), hipMemcpyDeviceToHost, s));
    GPMPC_HIP_CHECK(h, hipStreamSynchronize(s));
    return GPMPC_OK;
}

}  // namespace gpmpc_hip
