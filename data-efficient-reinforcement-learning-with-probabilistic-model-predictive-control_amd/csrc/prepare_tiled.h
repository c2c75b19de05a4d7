// prepare_tiled.h -- the large-memory (N >= 640) kernels of gpmpc_prepare: LDS-tiled products on the fp64 matrix cores
// (64 x 64 tiles of round 2a, 128 x 128 tiles with buffer-load operands of round 2b), the LDS-resident factorisation of a
// whole 128-column outer panel, and the batched product behind the recursive-doubling triangular inverse.  Drivers and the
// panel-level kernels are in prepare.hip; DESIGN.md 4.2 has the measurements.
#pragma once
#include "gpmpc_internal.h"

namespace gpmpc_hip {

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int NB = 32;   // panel width
constexpr int kTPadRows = 72;   // zero rows after every T_a (= kTPad of rollout_kernel.h)

// 1 / sqrt(d), d > 0: fp32 v_rsq seed + two Newton steps (6 dependent fp64 operations instead of ~35 for sqrt + divide)
__device__ inline double inv_sqrt_pos_p(double d) {
    double y = (double)__builtin_amdgcn_rsqf((float)d);
    const double h = 0.5 * d;
    y = y * fma(-h * y, y, 1.5);
    y = y * fma(-h * y, y, 1.5);
    return y;
}

// ------------------------------------------------------------------------------------------
// LDS-tiled symmetric rank-k products on the matrix cores (large N).  The round-1 kernels above give every wavefront one
// 16 x 16 tile with operands straight from L2: 2 loads per lane per MFMA, the A operand as 16 scattered 32-byte pieces --
// they ran at 7-21 TFLOP/s of the 78.6 TFLOP/s fp64 matrix peak.  Here a workgroup of 4 wavefronts owns a 64 x 64 tile of
// the result; 32-deep slices of both operands are staged through LDS with coalesced loads (256-byte row segments) and
// every wavefront forms a 32 x 32 sub-tile (2 x 2 MFMA tiles: 4 LDS reads per 4 MFMAs).  LDS layouts are chosen per
// operand orientation so that the fragment reads are conflict-free: rows x k with a stride of 34 doubles when the
// operand is stored row-major along k, k x columns with a stride of 80 doubles when it is stored along the columns.
constexpr int TS = 64;            // tile edge
constexpr int KC = 32;            // k-slice
constexpr int SI = KC + 2;        // LDS stride, i-major tiles (rows x k)
constexpr int SK = TS + 16;       // LDS stride, k-major tiles (k x columns)

// Trailing update of the outer-blocked Cholesky: C[i][j] -= sum_{p < w} L[i][k0 + p] L[j][k0 + p] for the lower triangle of
// rows / columns >= r0 (= k0 + w).  Both operands are rows of L: i-major staging.
__global__ __launch_bounds__(256) void syrk_outer_kernel(double* __restrict__ Kall, int N, int k0, int w) {
    __shared__ double As[TS * SI];
    __shared__ double Bs[TS * SI];
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (tj > ti) return;
    double* K = Kall + (size_t)blockIdx.z * N * N;
    const int r0 = k0 + w;
    const int i0 = r0 + ti * TS, j0 = r0 + tj * TS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
    // staging: element e = 256 u + tid of the 64 x 32 slice, row e / 32, k = e % 32 -- a wavefront moves two 256-byte row
    // segments per instruction and writes 64 consecutive doubles (+ one row skip) of LDS: coalesced and conflict-free
    const int srow = tid >> 5, sk = tid & 31;
    d4 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = {0.0, 0.0, 0.0, 0.0};
    double av[8], bv[8];
    auto fetch = [&](int p0) {
        const bool kin = (p0 + sk < w);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int row = 8 * u + srow;
            const int ra = (i0 + row < N) ? i0 + row : N - 1, rb = (j0 + row < N) ? j0 + row : N - 1;
            av[u] = kin ? K[(size_t)ra * N + k0 + p0 + sk] : 0.0;
            bv[u] = kin ? K[(size_t)rb * N + k0 + p0 + sk] : 0.0;
        }
    };
    fetch(0);
    for (int p0 = 0; p0 < w; p0 += KC) {
        __syncthreads();                                             // previous slice consumed
#pragma unroll
        for (int u = 0; u < 8; ++u) { As[(8 * u + srow) * SI + sk] = av[u]; Bs[(8 * u + srow) * SI + sk] = bv[u]; }
        __syncthreads();
        if (p0 + KC < w) fetch(p0 + KC);                             // next slice travels while this one is multiplied
#pragma unroll
        for (int kk = 0; kk < KC; kk += 4) {
            const double a0 = As[(wi + li) * SI + kk + lk], a1 = As[(wi + 16 + li) * SI + kk + lk];
            const double b0 = Bs[(wj + li) * SI + kk + lk], b1 = Bs[(wj + 16 + li) * SI + kk + lk];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + wi + 16 * x + lk + 4 * r, col = j0 + wj + 16 * y + li;
                if (row < N && col <= row) K[(size_t)row * N + col] -= acc[x][y][r];
            }
}

// iK = Y^T Y (Y = L^-1 lower triangular: Y[p][c] = 0 for p < c) on 64 x 64 tiles of the lower triangle, mirrored on store,
// plus T = beta beta^T - iK (upper triangle, diagonal halved).  Both operands are columns of Y: k-major staging.
__global__ __launch_bounds__(256) void syrk_inverse_tiled_kernel(const double* __restrict__ Yall, const double* __restrict__ beta,
                                                                 int N, double* __restrict__ iKall, double* __restrict__ Tall) {
    __shared__ double As[KC * SK];
    __shared__ double Bs[KC * SK];
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (tj > ti) return;
    const int a = blockIdx.z;
    const double* Y = Yall + (size_t)a * N * N;
    double* iK = iKall + (size_t)a * N * N;
    double* T = Tall + (size_t)a * (N + kTPadRows) * N;
    const double* be = beta + (size_t)a * N;
    const int i0 = ti * TS, j0 = tj * TS;                            // j0 <= i0
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
    // staging: element e = 256 u + tid of the 32 x 64 slice, k = e / 64, column e % 64 -- a wavefront moves one 512-byte
    // row segment per instruction and writes 64 consecutive doubles of LDS
    const int sp = tid >> 6, scol = tid & 63;
    d4 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = {0.0, 0.0, 0.0, 0.0};
    double av[8], bv[8];
    auto fetch = [&](int p0) {
        const int ci = i0 + scol, cj = j0 + scol;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = p0 + 4 * u + sp;
            av[u] = (p < N && ci < N) ? Y[(size_t)p * N + ci] : 0.0;
            bv[u] = (p < N && cj < N) ? Y[(size_t)p * N + cj] : 0.0;
        }
    };
    fetch(i0);
    for (int p0 = i0; p0 < N; p0 += KC) {                            // rows p < i0 of the A columns are zero
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 8; ++u) { As[(4 * u + sp) * SK + scol] = av[u]; Bs[(4 * u + sp) * SK + scol] = bv[u]; }
        __syncthreads();
        if (p0 + KC < N) fetch(p0 + KC);                             // next slice travels while this one is multiplied
#pragma unroll
        for (int kk = 0; kk < KC; kk += 4) {
            const double a0 = As[(kk + lk) * SK + wi + li], a1 = As[(kk + lk) * SK + wi + 16 + li];
            const double b0 = Bs[(kk + lk) * SK + wj + li], b1 = Bs[(kk + lk) * SK + wj + 16 + li];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + wi + 16 * x + lk + 4 * r, col = j0 + wj + 16 * y + li;
                if (row < N && col < N && col <= row) {
                    const double v = acc[x][y][r];
                    double t = be[row] * be[col] - v;
                    if (row == col) t *= 0.5;
                    iK[(size_t)row * N + col] = v;
                    T[(size_t)col * N + row] = t;                    // upper triangle only; the rest stays zero
                    if (row != col) iK[(size_t)col * N + row] = v;
                }
            }
}

// ---- 128 x 128 tiles, 8 wavefronts -------------------------------------------------------------------------------------
// A 64 x 64 tile moves 16 bytes per 16 multiply-adds through the L2: at N = 4096, D = 16 the three products above pull
// ~3.7 TB/s and sit at 37-40 % of the matrix peak with the matrix pipe idle 60 % of the time.  128 x 128 tiles halve the
// bytes per flop: 8 wavefronts, each a 32 x 64 sub-tile (2 x 4 MFMA tiles: 6 LDS reads per 8 MFMAs), 32-deep slices, the
// next slice travelling in registers while the current one is multiplied (64 MFMAs = 4096 matrix-pipe cycles per wavefront
// and slice against one pair of barriers).  Addresses: one uniform base pointer per operand advanced by scalar adds plus
// per-thread 32-bit offsets computed once; full slices are loaded unmasked, only the last one compares.
constexpr int T2 = 128;
constexpr int SI2 = KC + 2;        // i-major slice: 128 rows x 32 k, stride 34
constexpr int SK2 = T2 + 16;       // k-major slice: 32 k x 128 columns, stride 144 (144 mod 32 = 16: see SK)

// 1-D grid of nbatch x ntile workgroups -> (batch, tile).  Workgroup ids are dealt round-robin to the 8 XCDs; when the
// number of batches (GPs) is a multiple of 8 every XCD works through whole batches (a, a + 8, ...) tile by tile, so the
// workgroups sharing an L2 read the same operand blocks.  Otherwise (D = 4 at config 4) that would leave XCDs without work:
// batches are then laid out one after the other and their tiles spread over all XCDs.
__device__ inline void xcd_batch_tile(int id, int nbatch, int ntile, int& batch, int& tile) {
    if ((nbatch & 7) == 0) {
        const int x = id & 7, l = id >> 3;
        batch = x + 8 * (l / ntile);
        tile = l - (l / ntile) * ntile;
    } else {
        batch = id / ntile;
        tile = id - batch * ntile;
    }
}

// lower-triangle tile (ti >= tj) number t -> (ti, tj), row by row
__device__ inline void tri_tile(int t, int& ti, int& tj) {
    int r = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((r + 1) * (r + 2) / 2 <= t) ++r;
    while (r * (r + 1) / 2 > t) --r;
    ti = r; tj = t - r * (r + 1) / 2;
}

// Operand slices are fetched with raw buffer loads: one 32-bit byte offset per thread and operand, everything uniform
// (slice row, slice advance) in the scalar offset, and the range check of the descriptor returns zero past `bytes` -- rows
// past the end of a matrix need neither clamping nor masking.
__device__ inline __amdgpu_buffer_rsrc_t operand_rsrc(const double* p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)(unsigned)(bytes < 0xFFFFFFFFull ? bytes : 0xFFFFFFFFull), 0x00020000);
}
__device__ inline double operand_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
}

// 8 k-steps of one slice: As / Bs fragment base pointers already include the wavefront and lane offsets
template <int ASTEP, int AROW, int BSTEP, int BROW, bool FENCE = false>
__device__ inline void slice_mfma_2x4(d4 (&acc)[2][4], const double* __restrict__ Af, const double* __restrict__ Bf) {
#pragma unroll
    for (int kk = 0; kk < KC; kk += 4) {
        const double a0 = Af[kk * ASTEP], a1 = Af[kk * ASTEP + 16 * AROW];
        const double b0 = Bf[kk * BSTEP], b1 = Bf[kk * BSTEP + 16 * BROW], b2 = Bf[kk * BSTEP + 32 * BROW], b3 = Bf[kk * BSTEP + 48 * BROW];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[0][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b2, acc[0][2], 0, 0, 0);
        acc[0][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b3, acc[0][3], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        acc[1][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b2, acc[1][2], 0, 0, 0);
        acc[1][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b3, acc[1][3], 0, 0, 0);
        // FENCE: keep the fragment reads of the next k-step behind these MFMAs (12 fragment registers instead of 24; with
        // four wavefronts per SIMD the others cover the LDS latency)
        if (FENCE) __builtin_amdgcn_sched_barrier(0);
    }
}

// iK = Y^T Y and T = beta beta^T - iK as in syrk_inverse_tiled_kernel, 128 x 128 tiles of the lower triangle.  1-D grid:
// workgroup id -> XCD id & 7 (round-robin dispatch); every XCD works through whole GPs (a, a + 8, ...) tile row by tile row,
// so that the workgroups sharing its L2 read the same column blocks of the same Y.
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void syrk_inverse_t128_kernel(const double* __restrict__ Yall, const double* __restrict__ beta,
                                                                int N, int D, int ntile, double* __restrict__ iKall,
                                                                double* __restrict__ Tall) {
    __shared__ double As[KC * SK2];
    __shared__ double Bs[KC * SK2];
    int a, t;
    {
        xcd_batch_tile(blockIdx.x, D, ntile, a, t);
        if (a >= D) return;
    }
    int ti, tj;
    tri_tile(t, ti, tj);
    const double* Y = Yall + (size_t)a * N * N;
    double* iK = iKall + (size_t)a * N * N;
    double* T = Tall + (size_t)a * (N + kTPadRows) * N;
    const double* be = beta + (size_t)a * N;
    const int i0 = ti * T2, j0 = tj * T2;                            // j0 <= i0
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 64;
    // staging: element e = 512 u + tid of the 32 x 128 slice, k = e / 128 = 4 u + sp, column e % 128
    const int sp = tid >> 7, scol = tid & 127;
    const unsigned ca = (unsigned)((i0 + scol < N) ? i0 + scol : N - 1), cb = (unsigned)((j0 + scol < N) ? j0 + scol : N - 1);
    const unsigned offa = ((unsigned)sp * (unsigned)N + ca) * 8u, offb = ((unsigned)sp * (unsigned)N + cb) * 8u;
    const __amdgpu_buffer_rsrc_t rs = operand_rsrc(Y, (size_t)N * N * 8);      // rows p >= N read as zero
    d4 acc[2][4];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] = {0.0, 0.0, 0.0, 0.0};
    double av[8], bv[8];
    auto fetch = [&](int p0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned so = (unsigned)(p0 + 4 * u) * (unsigned)N * 8u;
            av[u] = operand_load(rs, offa, so);
            bv[u] = operand_load(rs, offb, so);
        }
    };
    const double* Af = As + lk * SK2 + wi + li;
    const double* Bf = Bs + lk * SK2 + wj + li;
    double* Aw = As + sp * SK2 + scol;
    double* Bw = Bs + sp * SK2 + scol;
    fetch(i0);
    for (int p0 = i0; p0 < N; p0 += KC) {                            // rows p < i0 of the A columns are zero
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 8; ++u) { Aw[4 * u * SK2] = av[u]; Bw[4 * u * SK2] = bv[u]; }
        __syncthreads();
        if (p0 + KC < N) fetch(p0 + KC);
        slice_mfma_2x4<SK2, 1, SK2, 1>(acc, Af, Bf);
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + wi + 16 * x + lk + 4 * r, col = j0 + wj + 16 * y + li;
                if (row < N && col < N && col <= row) {
                    const double v = acc[x][y][r];
                    double tt = be[row] * be[col] - v;
                    if (row == col) tt *= 0.5;
                    iK[(size_t)row * N + col] = v;
                    T[(size_t)col * N + row] = tt;                   // upper triangle only; the rest stays zero
                    if (row != col) iK[(size_t)col * N + row] = v;
                }
            }
}

// Trailing update of the outer-blocked Cholesky as in syrk_outer_kernel, 128 x 128 tiles of the lower triangle (w is a
// multiple of 32).  Same 1-D grid / XCD mapping as syrk_inverse_t128_kernel.
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void syrk_outer_t128_kernel(double* __restrict__ Kall, int N, int D, int ntile, int k0, int w, int nto) {
    __shared__ double S[2 * T2 * SI2];                               // A slice | B slice: one base register, constant offsets
    double* const As = S;
    double* const Bs = S + T2 * SI2;
    int a, t;
    {
        xcd_batch_tile(blockIdx.x, D, ntile, a, t);
        if (a >= D) return;
    }
    int ti, tj;
    // tiles of the first columns of the trailing matrix, column by column: column tj has rows tj .. nto - 1
    tj = 0;
    while (t >= nto - tj) { t -= nto - tj; ++tj; }
    ti = tj + t;
    double* K = Kall + (size_t)a * N * N;
    const int r0 = k0 + w;
    const int i0 = r0 + ti * T2, j0 = r0 + tj * T2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 64;
    // staging: element e = 512 u + tid of the 128 x 32 slice, row e / 32 = 16 u + srow, k = e % 32
    const int srow = tid >> 5, sk = tid & 31;
    const unsigned offa = ((unsigned)(i0 + srow) * (unsigned)N + (unsigned)sk) * 8u;
    const unsigned offb = ((unsigned)(j0 + srow) * (unsigned)N + (unsigned)sk) * 8u;
    const __amdgpu_buffer_rsrc_t rs = operand_rsrc(K, (size_t)N * N * 8);      // rows >= N read as zero
    d4 acc[2][4];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] = {0.0, 0.0, 0.0, 0.0};
    double av[8], bv[8];
    auto fetch = [&](int p0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned so = ((unsigned)(16 * u) * (unsigned)N + (unsigned)(k0 + p0)) * 8u;
            av[u] = operand_load(rs, offa, so);
            bv[u] = operand_load(rs, offb, so);
        }
    };
    const double* Af = As + (wi + li) * SI2 + lk;
    const double* Bf = Bs + (wj + li) * SI2 + lk;
    double* Aw = As + srow * SI2 + sk;
    fetch(0);
    for (int p0 = 0; p0 < w; p0 += KC) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 8; ++u) { Aw[16 * u * SI2] = av[u]; Aw[T2 * SI2 + 16 * u * SI2] = bv[u]; }
        __syncthreads();
        if (p0 + KC < w) fetch(p0 + KC);
        slice_mfma_2x4<1, SI2, 1, SI2, true>(acc, Af, Bf);
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + wi + 16 * x + lk + 4 * r, col = j0 + wj + 16 * y + li;
                if (row < N && col <= row) unsafeAtomicAdd(&K[(size_t)row * N + col], -acc[x][y][r]);   // one add per element and launch: no read latency to wait for
            }
}

// ---- whole outer panel in two launches ---------------------------------------------------------------------------------
// The 32-column panel chain (block factorisation + panel solve, 8 dependent launches per 128 columns) is ~150 us of
// critical path per outer panel at N = 4096 -- 18 % of the factorisation.  potrf_block128_kernel factorises the whole
// 128 x 128 diagonal block of an outer panel in ONE workgroup per GP with the block resident in LDS (stride 130: the
// MFMA operand reads of 16 rows x 2 k-values fall on 32 distinct banks): per 32 columns a left-looking update on the matrix
// cores, the register pivot loop of potrf_diag_fast_kernel, the solve of the rows below with L11^-T; the diagonal blocks
// are replaced by their inverses on the way and a row-block recursion turns the LDS copy into Y_KK = L_KK^-1 (scratch W in
// the unused upper blocks).  trsm_outer_t128_kernel then solves all rows below the block with one tiled product,
// L[rows, K] = A[rows, K] Y_KK^T.
constexpr int kPS = 130;

__global__ __launch_bounds__(1024) void potrf_block128_kernel(double* __restrict__ Kall, double* __restrict__ Yall, int N, int J0,
                                                              int* __restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Bk = smem;                                       // (128, 130) the block; diagonal 32-blocks become their inverses
    double* Yt = Bk + T2 * kPS;                              // (32, 33) L11^-T of the current panel
    double* colb = Yt + NB * 33;                             // 2 x 64 current / next pivot column (+ identity column)
    double* sinv = colb + 4 * NB;                            // (32) 1 / L_kk
    const int a = blockIdx.x;
    double* K = Kall + (size_t)a * N * N;
    double* Y = Yall + (size_t)a * N * N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int c32 = tid >> 5, r32 = tid & 31;
    const int nr = (N - J0 < T2) ? (N - J0) : T2;
    {
        // all 16 loads of a thread in flight (clamped addresses + selects: with a branch per element the compiler emitted
        // load - wait - store sixteen times, ~20 us of a 59 us kernel)
        const int c = tid & 127, r0 = tid >> 7;
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int r = r0 + 8 * u;
            const int rr = r < nr ? r : nr - 1, cc = c <= rr ? c : rr;
            v[u] = K[(size_t)(J0 + rr) * N + J0 + cc];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int r = r0 + 8 * u;
            Bk[r * kPS + c] = (r < nr && c <= r) ? v[u] : 0.0;
        }
    }
    __syncthreads();
    for (int k0 = 0; k0 < nr; k0 += NB) {
        const int nb = (nr - k0 < NB) ? (nr - k0) : NB;
        // (a) panel -= L[k0:, 0:k0] L[k0:k0+32, 0:k0]^T
        if (k0 > 0) {
            const int nrt = (nr - k0 + 15) >> 4;
            for (int t = wave; t < 2 * nrt; t += 16) {
                const int i0 = k0 + (t >> 1) * 16, j0 = (t & 1) * 16;
                d4 acc = {0.0, 0.0, 0.0, 0.0};
                const double* Ar = Bk + (i0 + li) * kPS + lk;
                const double* Br = Bk + (k0 + j0 + li) * kPS + lk;
                for (int pp = 0; pp < k0; pp += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ar[pp], Br[pp], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) Bk[(i0 + lk + 4 * r) * kPS + k0 + j0 + li] -= acc[r];
            }
            __syncthreads();
        }
        // (b) 32 x 32 diagonal block: thread (row r32, column c32), identity under it (see potrf_diag_fast_kernel)
        double a0 = (r32 < nb && c32 < nb && c32 <= r32) ? Bk[(k0 + r32) * kPS + k0 + c32] : 0.0;
        double a1 = (r32 == c32) ? 1.0 : 0.0;
        if (c32 == 0) { colb[r32] = a0; colb[NB + r32] = a1; }
        if (tid == 0) {
            if (!(a0 > 0.0) && info[a] == 0) info[a] = J0 + k0 + 1;
            sinv[0] = inv_sqrt_pos_p(a0);
        }
        if (tid >= nb && tid < NB) sinv[tid] = 0.0;
        __syncthreads();
        for (int k = 0; k + 1 < nb; ++k) {
            if (2 * wave + 1 > k) {
                const double* cb = colb + (k & 1) * 2 * NB;
                double* cn = colb + ((k + 1) & 1) * 2 * NB;
                const double inv = sinv[k];
                if (c32 > k && c32 < nb) {
                    const double lc = cb[c32] * inv;
                    a0 = fma(-(cb[r32] * inv), lc, a0);
                    a1 = fma(-(cb[NB + r32] * inv), lc, a1);
                    if (c32 == k + 1) {
                        cn[r32] = a0;
                        cn[NB + r32] = a1;
                        if (r32 == k + 1) {
                            if (!(a0 > 0.0) && info[a] == 0) info[a] = J0 + k0 + k + 2;
                            sinv[k + 1] = inv_sqrt_pos_p(a0);
                        }
                    }
                }
            }
            __syncthreads();
        }
        {
            const double sc = sinv[c32];                     // 0 for c >= nb
            const double l = (c32 <= r32) ? a0 * sc : 0.0, y = (r32 <= c32) ? a1 * sc : 0.0;   // y = Y11[c32][r32]
            Yt[r32 * 33 + c32] = y;
            Bk[(k0 + c32) * kPS + k0 + r32] = y;             // the diagonal block now holds Y11 (row c32, column r32)
            if (r32 < nb && c32 <= r32) K[(size_t)(J0 + k0 + r32) * N + J0 + k0 + c32] = l;
            if (r32 < nb && c32 < nb) Y[(size_t)(J0 + k0 + c32) * N + J0 + k0 + r32] = y;
        }
        __syncthreads();
        // (c) rows below: L21 = A21 L11^-T (a wavefront owns 16 rows and both column halves: it reads before it writes)
        {
            const int M = nr - k0 - nb;
            const int nrt = (M + 15) >> 4;
            for (int t = wave; t < nrt; t += 16) {
                const int i0 = k0 + nb + t * 16;
                d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
                const double* Ar = Bk + (i0 + li) * kPS + k0 + lk;
#pragma unroll
                for (int kk = 0; kk < NB; kk += 4) {
                    const double av = Ar[kk];
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Yt[(kk + lk) * 33 + li], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Yt[(kk + lk) * 33 + 16 + li], acc1, 0, 0, 0);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // the lanes exchange rows through Bk
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = i0 + lk + 4 * r;
                    Bk[row * kPS + k0 + li] = acc0[r];
                    Bk[row * kPS + k0 + 16 + li] = acc1[r];
                    if (row < nr) {
                        K[(size_t)(J0 + row) * N + J0 + k0 + li] = acc0[r];
                        K[(size_t)(J0 + row) * N + J0 + k0 + 16 + li] = acc1[r];
                    }
                }
            }
        }
        __syncthreads();
    }
    // Y_KK by row blocks, in place: Y[k, c] = -Y_kk (L[k, c:k] Y[c:k, c]); W (32 x k0) in rows 0..31, columns 32.. of Bk
    for (int k0 = NB; k0 < nr; k0 += NB) {
        const int nb = (nr - k0 < NB) ? (nr - k0) : NB;
        const int nct = k0 >> 4;
        for (int t = wave; t < 2 * nct; t += 16) {
            const int i0 = (t & 1) * 16, c0 = (t >> 1) * 16;
            d4 acc = {0.0, 0.0, 0.0, 0.0};
            const double* Ar = Bk + (k0 + i0 + li) * kPS + lk;
            const double* Bc = Bk + lk * kPS + c0 + li;
            for (int pp = c0 & ~3; pp < k0; pp += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ar[pp], Bc[pp * kPS], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) Bk[(i0 + lk + 4 * r) * kPS + NB + c0 + li] = acc[r];
        }
        __syncthreads();
        for (int t = wave; t < 2 * nct; t += 16) {
            const int i0 = (t & 1) * 16, c0 = (t >> 1) * 16;
            d4 acc = {0.0, 0.0, 0.0, 0.0};
            const double* Ar = Bk + (k0 + i0 + li) * kPS + k0 + lk;
            const double* Bc = Bk + lk * kPS + NB + c0 + li;
#pragma unroll
            for (int m = 0; m < NB; m += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ar[m], Bc[m * kPS], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + lk + 4 * r;
                Bk[(k0 + row) * kPS + c0 + li] = -acc[r];
                if (row < nb) Y[(size_t)(J0 + k0 + row) * N + J0 + c0 + li] = -acc[r];
            }
        }
        __syncthreads();
    }
}

// L[rows >= J0 + 128, J0 : J0 + 128] = A[rows, J0 : J0 + 128] Y_KK^T on 128-row tiles (k = 128), in place: a workgroup has
// read all of its rows (the last slice is in LDS behind a barrier) before any wavefront stores.
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4)))
void trsm_outer_t128_kernel(double* __restrict__ Kall, const double* __restrict__ Yall, int N, int D, int ntile, int J0, int nk) {
    __shared__ double S[2 * T2 * SI2];
    double* const As = S;
    double* const Bs = S + T2 * SI2;
    int a, t;
    {
        xcd_batch_tile(blockIdx.x, D, ntile, a, t);
        if (a >= D) return;
    }
    double* K = Kall + (size_t)a * N * N;
    const double* Y = Yall + (size_t)a * N * N;
    const int i0 = J0 + T2 + t * T2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 64;
    const int srow = tid >> 5, sk = tid & 31;
    const unsigned offa = ((unsigned)(i0 + srow) * (unsigned)N + (unsigned)sk) * 8u;
    const unsigned offb = ((unsigned)(J0 + srow) * (unsigned)N + (unsigned)sk) * 8u;
    const __amdgpu_buffer_rsrc_t ra = operand_rsrc(K, (size_t)N * N * 8);
    const __amdgpu_buffer_rsrc_t rb = operand_rsrc(Y, (size_t)N * N * 8);
    d4 acc[2][4];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] = {0.0, 0.0, 0.0, 0.0};
    double av[8], bv[8];
    auto fetch = [&](int p0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned so = ((unsigned)(16 * u) * (unsigned)N + (unsigned)(J0 + p0)) * 8u;
            av[u] = operand_load(ra, offa, so);
            bv[u] = operand_load(rb, offb, so);
        }
    };
    const double* Af = As + (wi + li) * SI2 + lk;
    const double* Bf = Bs + (wj + li) * SI2 + lk;
    double* Aw = As + srow * SI2 + sk;
    fetch(0);
    for (int p0 = 0; p0 < nk; p0 += KC) {                            // nk = 128 (a multiple of 32; rows of Y_KK past the block: zero)
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 8; ++u) { Aw[16 * u * SI2] = av[u]; Aw[T2 * SI2 + 16 * u * SI2] = bv[u]; }
        __syncthreads();
        if (p0 + KC < nk) fetch(p0 + KC);
        slice_mfma_2x4<1, SI2, 1, SI2, true>(acc, Af, Bf);
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + wi + 16 * x + lk + 4 * r, col = J0 + wj + 16 * y + li;
                if (row < N && col < N) K[(size_t)row * N + col] = acc[x][y][r];
            }
}

// Batched C = alpha A B on 128 x 128 tiles: A (M x Kd) row-major (i-major staging), B (Kd x NC) row-major (k-major staging).
// Batch z = a * nq + q: operand blocks at base + a * s? + q * q?; the last q has M_last (> 0) rows, the others M rows.
// kskip: B is lower triangular (k starts at the column tile); ktri: A is lower triangular (k ends with the row tile).
// 1-D grid with the XCD mapping of the kernels above (batches are dealt to the XCDs).
struct GemmBatch {
    const double* A; int lda; size_t sa, qa;
    const double* B; int ldb; size_t sb, qb;
    double* C; int ldc; size_t sc, qc;
    size_t a_rem;                     // elements from A (batch q = 0) to the end of its matrix
    int nq, nbatch, M, M_last, NC, Kd, kd_is_m, tiles_x, tiles;
    double alpha; int kskip, ktri;
};

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_nn_t128_kernel(GemmBatch g) {
    __shared__ double As[T2 * SI2];
    __shared__ double Bs[KC * SK2];
    int z, t;
    {
        xcd_batch_tile(blockIdx.x, g.nbatch, g.tiles, z, t);
        if (z >= g.nbatch) return;
    }
    const int a = z / g.nq, q = z - a * g.nq;
    const int M = (q == g.nq - 1) ? g.M_last : g.M;
    const int NC = g.NC;
    const int Kd = g.kd_is_m ? M : g.Kd;
    // long tiles first: with a triangular left factor (ktri) the k range grows with the row tile -> rows in descending order;
    // with a triangular right factor (kskip) it shrinks with the column tile -> column by column
    const int trows = g.tiles / g.tiles_x;
    int trow, tcol;
    if (g.kskip) { tcol = t / trows; trow = t - tcol * trows; }
    else { trow = g.ktri ? (trows - 1 - t / g.tiles_x) : t / g.tiles_x; tcol = t % g.tiles_x; }
    const int i0 = trow * T2, c0 = tcol * T2;
    if (i0 >= M) return;
    const double* A = g.A + (size_t)a * g.sa + (size_t)q * g.qa;
    const double* B = g.B + (size_t)a * g.sb + (size_t)q * g.qb;
    double* C = g.C + (size_t)a * g.sc + (size_t)q * g.qc;
    const int lda = g.lda, ldb = g.ldb;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 64;
    const int srow = tid >> 5, sk = tid & 31;                        // A slice 128 x 32
    const int sp = tid >> 7, scol = tid & 127;                       // B slice 32 x 128
    const int kbeg = g.kskip ? c0 : 0;
    int kend = Kd;
    if (g.ktri && i0 + T2 < kend) kend = i0 + T2;
    const unsigned offa = ((unsigned)(i0 + srow) * (unsigned)lda + (unsigned)sk) * 8u;
    const unsigned offb = ((unsigned)sp * (unsigned)ldb + (unsigned)(c0 + scol < NC ? c0 + scol : NC - 1)) * 8u;
    // A: rows past M are rows of the enclosing matrix (finite; their products are not stored) or, past its end, zero;
    // B: rows p >= kend read as zero, which also covers the k tail of A (whatever finite values it reads there)
    const __amdgpu_buffer_rsrc_t ra = operand_rsrc(A, (g.a_rem - (size_t)q * g.qa) * 8);
    const __amdgpu_buffer_rsrc_t rb = operand_rsrc(B, (size_t)kend * ldb * 8);
    d4 acc[2][4];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] = {0.0, 0.0, 0.0, 0.0};
    double av[8], bv[8];
    auto fetch = [&](int p0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            av[u] = operand_load(ra, offa, ((unsigned)(16 * u) * (unsigned)lda + (unsigned)p0) * 8u);
            bv[u] = operand_load(rb, offb, (unsigned)(p0 + 4 * u) * (unsigned)ldb * 8u);
        }
    };
    const double* Af = As + (wi + li) * SI2 + lk;
    const double* Bf = Bs + lk * SK2 + wj + li;
    double* Aw = As + srow * SI2 + sk;
    double* Bw = Bs + sp * SK2 + scol;
    if (kbeg < kend) fetch(kbeg);
    for (int p0 = kbeg; p0 < kend; p0 += KC) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 8; ++u) { Aw[16 * u * SI2] = av[u]; Bw[4 * u * SK2] = bv[u]; }
        __syncthreads();
        if (p0 + KC < kend) fetch(p0 + KC);
        slice_mfma_2x4<1, SI2, SK2, 1>(acc, Af, Bf);
    }
    const double alpha = g.alpha;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + wi + 16 * x + lk + 4 * r, col = c0 + wj + 16 * y + li;
                if (row < M && col < NC) C[(size_t)row * g.ldc + col] = alpha * acc[x][y][r];
            }
}

// C (M x NC) = alpha * A (M x Kd, row-major, lda) * B (Kd x NC, row-major, ldb) on 64 x 64 tiles, batched over blockIdx.z.
// A rows are staged i-major, B rows k-major (see above).  kskip: B[p][c] = 0 for p < c (a lower-triangular right factor), so
// the k range of column tile c0 starts at c0; ktri: A[i][p] = 0 for p > i (a lower-triangular left factor), so it ends at
// i0 + 64.  Used for the triangular inverse by 128-row blocks: W = L[K, c:K] Y[c:K, c], then Y[K, c] = -Y_KK W.
__global__ __launch_bounds__(256) void gemm_nn_tiled_kernel(const double* __restrict__ Aall, int lda, size_t sa,
                                                            const double* __restrict__ Ball, int ldb, size_t sb,
                                                            double* __restrict__ Call, int ldc, size_t sc, int M, int NC, int Kd,
                                                            double alpha, int kskip, int ktri) {
    __shared__ double As[TS * SI];
    __shared__ double Bs[KC * SK];
    const double* A = Aall + (size_t)blockIdx.z * sa;
    const double* B = Ball + (size_t)blockIdx.z * sb;
    double* C = Call + (size_t)blockIdx.z * sc;
    const int i0 = blockIdx.y * TS, c0 = blockIdx.x * TS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
    const int srow = tid >> 5, sk = tid & 31;                        // A slice 64 x 32: element 256 u + tid -> row 8 u + srow, k = sk
    const int sp = tid >> 6, scol = tid & 63;                        // B slice 32 x 64: element 256 u + tid -> k = 4 u + sp, column scol
    int kbeg = kskip ? (c0 & ~(KC - 1)) : 0;
    int kend = Kd;
    if (ktri && i0 + TS < kend) kend = i0 + TS;
    d4 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = {0.0, 0.0, 0.0, 0.0};
    double av[8], bv[8];
    auto fetch = [&](int p0) {
        const int pa = p0 + sk, cb = c0 + scol;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int row = 8 * u + srow;
            const int ra = (i0 + row < M) ? i0 + row : M - 1;
            av[u] = (pa < kend) ? A[(size_t)ra * lda + pa] : 0.0;
            const int pb = p0 + 4 * u + sp;
            bv[u] = (pb < kend && cb < NC) ? B[(size_t)pb * ldb + cb] : 0.0;
        }
    };
    if (kbeg < kend) fetch(kbeg);
    for (int p0 = kbeg; p0 < kend; p0 += KC) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 8; ++u) { As[(8 * u + srow) * SI + sk] = av[u]; Bs[(4 * u + sp) * SK + scol] = bv[u]; }
        __syncthreads();
        if (p0 + KC < kend) fetch(p0 + KC);
#pragma unroll
        for (int kk = 0; kk < KC; kk += 4) {
            const double a0 = As[(wi + li) * SI + kk + lk], a1 = As[(wi + 16 + li) * SI + kk + lk];
            const double b0 = Bs[(kk + lk) * SK + wj + li], b1 = Bs[(kk + lk) * SK + wj + 16 + li];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + wi + 16 * x + lk + 4 * r, col = c0 + wj + 16 * y + li;
                if (row < M && col < NC) C[(size_t)row * ldc + col] = alpha * acc[x][y][r];
            }
}

}  // namespace gpmpc_hip
