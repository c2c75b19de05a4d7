// prepare_small.hip -- the whole once-per-control-step factorisation in ONE launch for small memories (N <= 256).
//
// Same results as the panel-by-panel path of prepare.hip (reference rl_gp_mpc/control_objects/models/gp_model.py:400-431,
// 182-191): K_a = s2_a exp(-1/2 |x - x'|^2_l) + noise_a I, L_a = chol(K_a), Y_a = L_a^-1, beta_a = Y^T (Y y),
// iK_a = Y^T Y, T_a = beta beta^T - iK (upper triangle, diagonal halved), plus the packed inputs (X^T, 1/l^2, data range)
// and the record of what the factors were computed from.
//
// Why a second path: at config-2 size (N = 200, D = 3) the panel path is ~35 dependent launches of kernels that each
// fill a handful of CUs -- 0.49 ms, half of a control step, all of it launch latency and dependent-chain latency.
// Here one workgroup of 1024 threads owns one GP from the Gram matrix to T_a; the matrices (<= 512 KiB) live in L2,
// every phase hand-off is a workgroup barrier.  What bounds it is the length of the dependent fp64 chains (a
// dependent v_fma_f64 issues every ~40 cycles on this part), so the pivot loop is arranged for a short chain:
//   * 1/sqrt(pivot) from an fp32 v_rsq seed + two Newton steps (6 dependent fp64 ops; sqrt + divide would be ~35),
//     computed by the thread that owns the next pivot right after its own update -- ONE barrier per pivot;
//   * columns are never divided: every update multiplies by the two scaled factors, L is scaled on the way out;
//   * the panel solve is a product with the inverted diagonal block on the matrix cores, not a substitution.
// The N^3 parts (trailing update, L^-1 by row blocks, Y^T Y) are 16 x 16 fp64 MFMA tiles dealt to the 16 wavefronts.
#include "gpmpc_internal.h"

namespace gpmpc_hip {

typedef double sd4 __attribute__((ext_vector_type(4)));
constexpr int kSB = 32;            // panel width
constexpr int kSTPad = 72;         // zero rows after every T_a (= kTPad of rollout_kernel.h)
constexpr int kSmallMaxN = 256;

struct SmallPrepArgs {
    const double *X, *Y, *ls, *os, *noise;
    int N, D, E;
    double *Xt, *ils2, *var, *logvar, *xrange;      // packed inputs (written by workgroup 0)
    double *Xc, *Yc, *hyp;                          // record of (X, Y, hyper-parameters) for the reuse test of the next call
    double *K, *Yinv, *z, *beta, *iK, *T;
    int* info;
    int cholesky_only;       // 1: stop after the factorisation (L, inverted diagonal blocks); the N^3 products that follow
                             // (L^-1 by row blocks, beta, Y^T Y) then run as wide launches of prepare.hip's kernels
};

// LDS hand-off between lanes of ONE wavefront: LDS operations of a wave complete in issue order; the fences keep the
// compiler from moving accesses across
__device__ inline void wave_lds_sync_s() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 1 / sqrt(d), d > 0: fp32 seed (1 ulp) + two Newton steps y <- y (1.5 - 0.5 d y^2): relative error ~2^-85 before rounding
__device__ inline double inv_sqrt_pos(double d) {
    double y = (double)__builtin_amdgcn_rsqf((float)d);
    const double h = 0.5 * d;
    y = y * fma(-h * y, y, 1.5);
    y = y * fma(-h * y, y, 1.5);
    return y;
}

// acc += sum over p in [pbeg, pend) of A(p) B(p) for one 16 x 16 tile, operands straight from global memory (L2): the
// loads of U k-steps are issued before the first MFMA of the group, so a group costs one L2 round trip, not U.
template <int U, typename FA, typename FB>
__device__ inline void mfma_kloop(sd4& acc, int pbeg, int pend, int lk, FA loadA, FB loadB) {
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

__global__ __launch_bounds__(1024) void prepare_small_kernel(const SmallPrepArgs p) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int a = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = p.N, D = p.D, E = p.E;
    const int ES = E | 1;                                   // odd row stride of the scaled inputs: conflict-free column reads
    double* xs = sm;                                        // (N, ES) inputs scaled by 1 / l_a
    double* Lp = sm;                                        // P2: (N - k0 + 32, 33) panel + 32 identity rows (xs is dead by then)
    double* colb = Lp + (kSmallMaxN + kSB) * 33;            // 2 x (N + 32) current / next pivot column
    double* sinv = colb + 2 * (kSmallMaxN + kSB);           // (32) 1 / L_kk of the panel
    double* Yt = sinv + kSB;                                // (32, 33) L11^-T of the panel
    double* K = p.K + (size_t)a * N * N;
    double* Yv = p.Yinv + (size_t)a * N * N;
    double* iK = p.iK + (size_t)a * N * N;
    double* T = p.T + (size_t)a * (N + kSTPad) * N;
    double* be = p.beta + (size_t)a * N;
    double* zv = p.z + (size_t)a * N;
#ifdef GPMPC_PROF_ON
    long long stamp[12];
    int nstamp = 0;
    long long t_potrf = 0, t_trsm = 0, t_trail = 0;
#define SMALL_STAMP() do { stamp[nstamp++] = __builtin_readcyclecounter(); } while (0)
    SMALL_STAMP();
#else
#define SMALL_STAMP() do { } while (0)
#endif

    // ---- P0: packed inputs, data range, state record (workgroup 0); scaled inputs of this GP -------------------------
    if (a == 0) {
        for (int idx = tid; idx < N * E; idx += 1024) {
            const int e = idx / N, pt = idx - e * N;
            p.Xt[idx] = p.X[(size_t)pt * E + e];
            p.Xc[idx] = p.X[idx];
        }
        for (int idx = tid; idx < N * D; idx += 1024) p.Yc[idx] = p.Y[idx];
        for (int idx = tid; idx < D * E; idx += 1024) { const double l = p.ls[idx]; p.ils2[idx] = 1.0 / (l * l); p.hyp[idx] = l; }
        if (tid < D) {
            p.var[tid] = p.os[tid]; p.logvar[tid] = log(p.os[tid]);
            p.hyp[(size_t)D * E + tid] = p.os[tid];
            p.hyp[(size_t)D * E + kMaxD + tid] = p.noise[tid];
        }
        for (int e = wave; e < E; e += 16) {                // per-dimension min / max: one wavefront per input dimension
            double lo = INFINITY, hi = -INFINITY;
            for (int pt = lane; pt < N; pt += 64) { const double v = p.X[(size_t)pt * E + e]; lo = fmin(lo, v); hi = fmax(hi, v); }
            for (int off = 32; off >= 1; off >>= 1) { lo = fmin(lo, __shfl_xor(lo, off, 64)); hi = fmax(hi, __shfl_xor(hi, off, 64)); }
            if (lane == 0) { p.xrange[e] = lo; p.xrange[E + e] = hi; }
        }
    }
    for (int idx = tid; idx < N * E; idx += 1024) {
        const int pt = idx / E, e = idx - pt * E;
        const double l = p.ls[a * E + e];
        xs[pt * ES + e] = p.X[idx] * sqrt(1.0 / (l * l));   // the same x * sqrt(1 / l^2) as gram_kernel of prepare.hip
    }
    if (tid == 0) p.info[a] = 0;
    __syncthreads();

    SMALL_STAMP();
    // ---- P1: Gram matrix, lower triangle (gp_model.py:425,427); Y := 0 ----------------------------------------------------
    {
        const double va = p.os[a], nz = p.noise[a];
        for (int i = wave; i < N; i += 16) {
            for (int j = lane; j <= i; j += 64) {
                double s = 0.0;
                for (int e = 0; e < E; ++e) { const double d = xs[i * ES + e] - xs[j * ES + e]; s = fma(d, d, s); }
                double v = va * exp(-0.5 * s);
                if (i == j) v += nz;
                K[(size_t)i * N + j] = v;
            }
        }
        for (int idx = tid; idx < N * N; idx += 1024) Yv[idx] = 0.0;
    }
    __syncthreads();

    SMALL_STAMP();
    // ---- P2: blocked LEFT-looking Cholesky, panel width 32 (gp_model.py:427) ------------------------------------------
    // Per panel: (a) panel -= L[k0:, :k0] L[k0:k0+32, :k0]^T on the matrix cores (operands streamed from L2, 8 k-steps of
    // loads in flight), result into LDS; (b) unblocked elimination of the 32 x 32 diagonal block in registers -- thread
    // (row rr, column c) owns element (rr, c), a wavefront owns two columns and drops out once both are done -- with the
    // pivot column handed on through LDS, one barrier per pivot.  The 32 x 32 identity rides along under the block: it
    // comes out as L11^-T, which is both the panel solve (L21 = A21 L11^-T, a product on the matrix cores with both
    // operands in LDS) and the inverse of the diagonal block the row-block recursion of P3 needs.
    const int c32 = tid >> 5, r32 = tid & 31;                // column-major over the wavefronts: wave w <-> columns 2w, 2w + 1
    const int li = lane & 15, lk = lane >> 4;
    for (int k0 = 0; k0 < N; k0 += kSB) {
        const int nb = (N - k0 < kSB) ? (N - k0) : kSB;
        const int nr = N - k0;                               // rows of the panel
#ifdef GPMPC_PROF_ON
        const long long tp0 = __builtin_readcyclecounter();
#endif
        // (a) updated panel -> Lp
        {
            const int nrt = (nr + 15) >> 4;
            for (int t = wave; t < nrt; t += 16) {
                const int i0 = k0 + t * 16;
                sd4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
                double c0v[4], c1v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {                // the Gram values of this tile (lower triangle only is defined)
                    const int row = i0 + lk + 4 * r;
                    const bool in0 = row < N && li < nb && k0 + li <= row, in1 = row < N && 16 + li < nb && k0 + 16 + li <= row;
                    c0v[r] = in0 ? K[(size_t)row * N + k0 + li] : 0.0;
                    c1v[r] = in1 ? K[(size_t)row * N + k0 + 16 + li] : 0.0;
                }
                if (k0 > 0) {
                    const int ra = (i0 + li < N) ? i0 + li : N - 1;
                    const int rb0 = (li < nb) ? k0 + li : k0, rb1 = (16 + li < nb) ? k0 + 16 + li : k0;
                    const double* Ar = K + (size_t)ra * N;
                    const double* B0 = K + (size_t)rb0 * N;
                    const double* B1 = K + (size_t)rb1 * N;
                    constexpr int U = 8;
                    for (int pp = 0; pp < k0; pp += 4 * U) {             // k0 is a multiple of 32 = 4 U
                        double av[U], b0[U], b1[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) { const int pk = pp + 4 * u + lk; av[u] = Ar[pk]; b0[u] = B0[pk]; b1[u] = B1[pk]; }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], b0[u], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], b1[u], acc1, 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int lr = t * 16 + lk + 4 * r;      // row inside the panel
                    Lp[lr * 33 + li] = c0v[r] - acc0[r];
                    Lp[lr * 33 + 16 + li] = c1v[r] - acc1[r];
                }
            }
        }
        __syncthreads();
#ifdef GPMPC_PROF_ON
        const long long tp1 = __builtin_readcyclecounter();
        t_potrf += tp1 - tp0;
#endif
        // (b) elimination of the diagonal block (element (rr, c)) and of the identity under it (element (32 + rr, c))
        double a0 = (r32 < nb && c32 < nb && c32 <= r32) ? Lp[r32 * 33 + c32] : 0.0;
        double a1 = (r32 == c32) ? 1.0 : 0.0;
        if (c32 == 0) { colb[r32] = a0; colb[kSB + r32] = a1; }
        if (tid == 0) {
            if (!(a0 > 0.0) && p.info[a] == 0) p.info[a] = k0 + 1;
            sinv[0] = inv_sqrt_pos(a0);
        }
        if (tid >= nb && tid < kSB) sinv[tid] = 0.0;
        __syncthreads();
        for (int k = 0; k + 1 < nb; ++k) {
            if (2 * wave + 1 > k) {                          // wave-uniform: both columns of a finished wave are final
                const double* cb = colb + (k & 1) * (kSmallMaxN + kSB);
                double* cn = colb + ((k + 1) & 1) * (kSmallMaxN + kSB);
                const double inv = sinv[k];
                if (c32 > k && c32 < nb) {
                    const double lc = cb[c32] * inv;
                    a0 = fma(-(cb[r32] * inv), lc, a0);
                    a1 = fma(-(cb[kSB + r32] * inv), lc, a1);
                    if (c32 == k + 1) {
                        cn[r32] = a0;
                        cn[kSB + r32] = a1;
                        if (r32 == k + 1) {
                            if (!(a0 > 0.0) && p.info[a] == 0) p.info[a] = k0 + k + 2;
                            sinv[k + 1] = inv_sqrt_pos(a0);
                        }
                    }
                }
            }
            __syncthreads();
        }
#ifdef GPMPC_PROF_ON
        const long long tp2 = __builtin_readcyclecounter();
        t_trsm += tp2 - tp1;
#endif
        // scale on the way out: L_rc = value / L_cc; the identity rows are L11^-T (row m, column c = Y11[c][m])
        {
            const double sc = sinv[c32];                     // 0 for c >= nb
            const double l = (c32 <= r32) ? a0 * sc : 0.0, y = (r32 <= c32) ? a1 * sc : 0.0;
            Yt[r32 * 33 + c32] = y;
            if (r32 < nb && c32 <= r32) K[(size_t)(k0 + r32) * N + k0 + c32] = l;
            if (r32 < nb && c32 < nb) Yv[(size_t)(k0 + c32) * N + k0 + r32] = y;
        }
        __syncthreads();
        // panel solve on the matrix cores: L21 = A21 L11^-T, A21 = rows 32.. of Lp, B[k][j] = Yt[k][j]
        {
            const int M = nr - nb;
            const int nrt = (M + 15) >> 4;
            for (int t = wave; t < 2 * nrt; t += 16) {
                const int i0 = (t >> 1) * 16, j0 = (t & 1) * 16;
                sd4 acc = {0.0, 0.0, 0.0, 0.0};
                const double* Ar = Lp + (size_t)(nb + i0 + li) * 33;
#pragma unroll
                for (int kk = 0; kk < kSB; kk += 4)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ar[kk + lk], Yt[(kk + lk) * 33 + j0 + li], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = i0 + lk + 4 * r, col = j0 + li;
                    if (row < M && col < nb) K[(size_t)(k0 + nb + row) * N + k0 + col] = acc[r];
                }
            }
        }
        __syncthreads();
#ifdef GPMPC_PROF_ON
        t_trail += __builtin_readcyclecounter() - tp2;
#endif
    }

    SMALL_STAMP();
    if (p.cholesky_only) {
        // T_a := 0 (its upper triangle is filled by the Y^T Y launch that follows): saves the memset launch
        for (int idx = tid; idx < (N + kSTPad) * N; idx += 1024) T[idx] = 0.0;
        return;
    }
    // ---- P3: Y = L^-1 by row blocks: Y[k, c] = -Ykk (L[k, :k] Y[:k, c]) for c < k ------------------------------------------
    for (int k0 = kSB; k0 < N; k0 += kSB) {
        const int nb = (N - k0 < kSB) ? (N - k0) : kSB;
        const int nct = k0 >> 4;                               // column tiles of 16 (k0 is a multiple of 32)
        // W = L[k, 0:k0] Y[0:k0, 0:k0] -> scratch in T_a (32 x k0, row stride k0)
        for (int t = wave; t < 2 * nct; t += 16) {
            const int i0 = (t & 1) * 16, c0 = (t >> 1) * 16;
            sd4 acc = {0.0, 0.0, 0.0, 0.0};
            const bool rowin = (i0 + li < nb);
            const double* Arow = K + (size_t)(k0 + (rowin ? i0 + li : 0)) * N;
            const double* Bcol = Yv + c0 + li;
            mfma_kloop<16>(acc, c0, k0, lk,                                             // Y[p][c] = 0 for p < c
                          [&](int pk) { return rowin ? Arow[pk] : 0.0; },
                          [&](int pk) { return Bcol[(size_t)pk * N]; });
#pragma unroll
            for (int r = 0; r < 4; ++r) T[(size_t)(i0 + lk + 4 * r) * k0 + c0 + li] = acc[r];
        }
        __syncthreads();
        // Y[k, c] = -Ykk W   (Ykk = the block's inverse of P2b; lower triangular)
        for (int t = wave; t < 2 * nct; t += 16) {
            const int i0 = (t & 1) * 16, c0 = (t >> 1) * 16;
            sd4 acc = {0.0, 0.0, 0.0, 0.0};
            const bool rowin = (i0 + li < nb);
            const double* Arow = Yv + (size_t)(k0 + (rowin ? i0 + li : 0)) * N + k0;
            const double* Bcol = T + c0 + li;
            mfma_kloop<8>(acc, 0, kSB, lk,
                          [&](int m) { return (rowin && m < nb) ? Arow[m] : 0.0; },
                          [&](int m) { return Bcol[(size_t)m * k0]; });
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + lk + 4 * r;
                if (row < nb) Yv[(size_t)(k0 + row) * N + c0 + li] = -acc[r];
            }
        }
        __syncthreads();
    }

    SMALL_STAMP();
    // ---- P4: z = Y y, beta = Y^T z (= cholesky_solve(y, L), gp_model.py:429-430) ------------------------------------------
    for (int row = wave; row < N; row += 16) {
        double s = 0.0;
        for (int q = lane; q <= row; q += 64) s = fma(Yv[(size_t)row * N + q], p.Y[(size_t)q * D + a], s);
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) zv[row] = s;
    }
    __syncthreads();
    {
        // four threads per column (rows q = i + part, i + part + 4, ...), partial sums combined in a fixed order;
        // loads of 8 rows in flight
        const int i = tid >> 2, part = tid & 3;
        double s = 0.0;
        if (i < N) {
            for (int q0 = i + part; q0 < N; q0 += 32) {
                double yv[8], zz[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int q = q0 + 4 * u;
                    yv[u] = (q < N) ? Yv[(size_t)q * N + i] : 0.0;
                    zz[u] = (q < N) ? zv[q] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) s = fma(yv[u], zz[u], s);
            }
        }
        const double s1 = __shfl_xor(s, 1, 64);
        const double pair = (part & 1) ? s1 + s : s + s1;
        const double s2 = __shfl_xor(pair, 2, 64);
        if (i < N && part == 0) be[i] = pair + s2;
    }
    __syncthreads();

    SMALL_STAMP();
    // ---- P5: iK = Y^T Y (gp_model.py:428), T = beta beta^T - iK (upper triangle, diagonal halved), zero elsewhere ----------
    for (int idx = tid; idx < (N + kSTPad) * N; idx += 1024) T[idx] = 0.0;
    __syncthreads();
    {
        const int nt = (N + 15) >> 4;
        const int ntri = nt * (nt + 1) / 2;
        for (int t = wave; t < ntri; t += 16) {
            int ti = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
            while (ti * (ti + 1) / 2 > t) --ti;
            while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
            const int tj = t - ti * (ti + 1) / 2;
            const int i0 = ti * 16, j0 = tj * 16;              // j0 <= i0
            sd4 acc = {0.0, 0.0, 0.0, 0.0};
            const int ci = (i0 + li < N) ? i0 + li : N - 1, cj = (j0 + li < N) ? j0 + li : N - 1;    // clamped: masked on store
            mfma_kloop<16>(acc, i0, N, lk,                                                             // Y[p][c] = 0 for p < c
                          [&](int pk) { return Yv[(size_t)pk * N + ci]; },
                          [&](int pk) { return Yv[(size_t)pk * N + cj]; });
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + lk + 4 * r, col = j0 + li;
                if (row < N && col < N && col <= row) {
                    const double v = acc[r];
                    double tv = be[row] * be[col] - v;
                    if (row == col) tv *= 0.5;
                    iK[(size_t)row * N + col] = v;
                    T[(size_t)col * N + row] = tv;
                    if (row != col) iK[(size_t)col * N + row] = v;
                }
            }
        }
    }
#ifdef GPMPC_PROF_ON
    __syncthreads();
    SMALL_STAMP();
    if (a == 0 && tid == 0) {
        printf("prepare_small N=%d: P0 %lld  P1 gram %lld  P2 chol %lld (panel update %lld, pivots %lld, solve + store %lld)  P3 trinv %lld  "
               "P4 beta %lld  P5 syrk %lld  total %lld cycles\n", N, stamp[1] - stamp[0], stamp[2] - stamp[1], stamp[3] - stamp[2],
               t_potrf, t_trsm, t_trail, stamp[4] - stamp[3], stamp[5] - stamp[4], stamp[6] - stamp[5], stamp[6] - stamp[0]);
    }
#endif
}

// Y = L^-1 below the diagonal blocks, ONE launch: workgroup (cb, a) owns the 32-column block cb of Y_a and walks down its
// row blocks, Y[k, cb] = -Y_kk (L[k, cb:k] Y[cb:k, cb]) -- each step only needs rows of its own column block, which it
// wrote itself, so there is no dependence between workgroups (the row-block form of prepare.hip needs one launch per
// row block: 6 dependent launches at N = 200).  4 wavefronts = the 2 x 2 tiles of a 32 x 32 block.
__global__ __launch_bounds__(256) void trinv_cols_small_kernel(const double* __restrict__ Lall, double* __restrict__ Yall, int N) {
    __shared__ double Ws[kSB * 33];
    const int a = blockIdx.y, c0 = blockIdx.x * kSB;
    const double* L = Lall + (size_t)a * N * N;
    double* Y = Yall + (size_t)a * N * N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int wi = (wave >> 1) * 16, wj = (wave & 1) * 16;
    for (int k0 = c0 + kSB; k0 < N; k0 += kSB) {
        const int nb = (N - k0 < kSB) ? (N - k0) : kSB;
        sd4 acc = {0.0, 0.0, 0.0, 0.0};
        const bool rowin = (wi + li < nb);
        const double* Ar = L + (size_t)(k0 + (rowin ? wi + li : 0)) * N;
        const double* Bc = Y + c0 + wj + li;                          // c0 + 31 < k0 <= N: always inside
        mfma_kloop<16>(acc, c0, k0, lk, [&](int pk) { return rowin ? Ar[pk] : 0.0; }, [&](int pk) { return Bc[(size_t)pk * N]; });
#pragma unroll
        for (int r = 0; r < 4; ++r) Ws[(wi + lk + 4 * r) * 33 + wj + li] = acc[r];
        __syncthreads();
        sd4 acc2 = {0.0, 0.0, 0.0, 0.0};
        const double* Yk = Y + (size_t)(k0 + (rowin ? wi + li : 0)) * N + k0;
#pragma unroll
        for (int kk = 0; kk < kSB; kk += 4) {
            const int m = kk + lk;
            const double av = (rowin && m < nb) ? Yk[m] : 0.0;
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Ws[m * 33 + wj + li], acc2, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wi + lk + 4 * r;
            if (row < nb) Y[(size_t)(k0 + row) * N + c0 + wj + li] = -acc2[r];
        }
        __syncthreads();                                             // rows k0.. of this column block are read by the next step
    }
}

// Host side: returns 1 when the fused path handled the whole call (N <= 256), 2 when it did the factorisation, the
// triangular inverse and the zero fill of T, and the caller has to run beta and Y^T Y (run_prepare's tail), 0 to fall
// through, < 0 on error.
int run_prepare_small(Handle* h, const double* X, const double* Y, const double* ls, const double* os, const double* noise,
                      int N, int D, int E, hipStream_t s) {
    // Crossover with the panel chain, measured (round 3, tools/gpu_prepare_bench.py; the LDS arrays were sized by N for the test):
    // N = 200: 0.217 vs 0.240 ms, 256: 0.344 vs 0.299, 300: 0.387 vs 0.337, 400: 0.84 vs 0.49, 500: 1.15 vs 0.60 -- one CU's
    // matrix cores do the N^3 / 3 of the panel updates, so the single launch loses once the chain's launches fill more than a CU.
    constexpr int kSmallPathMaxN = 240;
    if (N > kSmallPathMaxN || N < 1 || h->opt_fused_prepare == 0) return 0;
    SmallPrepArgs p;
    // One workgroup per GP is the right shape for the factorisation (a chain of 200 dependent pivots) but not for the
    // N^3 products after it: on one CU they are matrix-core-bound at ~50 k cycles each for N = 200.  From N = 96 up only
    // the factorisation stays in the single launch (option "fused_prepare" = 2 forces everything into it, 3 never).
    p.cholesky_only = (h->opt_fused_prepare == 3 || (h->opt_fused_prepare != 2 && N >= 96)) ? 1 : 0;
    p.X = X; p.Y = Y; p.ls = ls; p.os = os; p.noise = noise;
    p.N = N; p.D = D; p.E = E;
    p.Xt = h->Xt.p; p.ils2 = h->ils2.p; p.var = h->var.p; p.logvar = h->logvar.p; p.xrange = h->xrange.p;
    p.Xc = h->Xc.p; p.Yc = h->Yc.p; p.hyp = h->hyp.p;
    p.K = h->gram.p; p.Yinv = h->linv.p; p.z = h->zvec.p; p.beta = h->beta.p; p.iK = h->iK.p; p.T = h->Tm.p;
    p.info = h->info;
    const int ES = E | 1;
    const size_t lds_a = (size_t)((N * ES + 1) & ~1);                                          // P1: scaled inputs
    const size_t lds_b = (size_t)(kSmallMaxN + kSB) * 33 + 2 * (kSmallMaxN + kSB) + kSB + kSB * 33;   // P2: panel, pivot columns, 1 / L_kk, L11^-T
    const size_t lds = (lds_a > lds_b ? lds_a : lds_b) * sizeof(double);
    const void* kern = reinterpret_cast<const void*>(prepare_small_kernel);
    if (lds > 64 * 1024) {
        int rc = allow_full_lds(h, kern);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(prepare_small_kernel, dim3(D), dim3(1024), lds, s, p);
    if (p.cholesky_only && N > kSB)
        hipLaunchKernelGGL(trinv_cols_small_kernel, dim3((N + kSB - 1) / kSB - 1, D), dim3(256), 0, s, h->gram.p, h->linv.p, N);
    GPMPC_HIP_CHECK(h, hipGetLastError());
    return p.cholesky_only ? 2 : 1;
}

}  // namespace gpmpc_hip
