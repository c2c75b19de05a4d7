// rollout_stream_kernel.h -- large-N variant of the GP-MPC inner loop for gfx950 (MI355X, CDNA4).
//
// Same mathematics, same C ABI entry and the same item loops as rollout_kernel.h, for memory sizes whose
// per-point arrays (nu, lb, row records: O(N (2D + G(D+2))) doubles) no longer fit the 160 KiB LDS of a
// CU (config 5: N = 4096, D = 16).  Nothing per-point is materialised:
//   * mean part, one output dimension a at a time: every thread streams its points (coalesced reads of
//     X^T), keeps the D + 1 partial sums  sum lb_a [1, nu]  in registers, one block reduction per a;
//   * pairs (a, b) one at a time: the N column factors go to LDS (8 N bytes); the row records
//     {ea_i | ka'_i, ra_i | beta_ai, g_i} are produced on the fly for one chunk of <= 64 rows at a time
//     into a double-buffered LDS stage (the chunk after next is filled while the current one is consumed:
//     one workgroup barrier per chunk); each wave owns the 64-column blocks w, w + NW, ... and keeps ONE
//     accumulator across all row chunks, so a pair costs a single wavefront reduction.
// Row operands are therefore LDS broadcasts exactly as in the LDS-resident kernel (the first round-1
// version of this variant re-read the records from L2 for every wave and ran ~7x off its VALU bound).
//
// Pairwise pass on the fp64 matrix cores (DP = 8, 16).  At D = 16 a row record is 18 doubles; broadcasting it from LDS to every
// wave for every row (the layout of the LDS-resident kernel) made the round-1 version of this kernel LDS-bound at 0.3 of its
// VALU roofline.  The exponent's bilinear part  c_ij = g_i . w_j  (inner dimension D) is GEMM-shaped at this size: a
// 16 x 16 tile of c is DP/4 v_mfma_f64_16x16x4_f64 -- A operand g_i straight from the LDS stage (one conflict-free
// ds_read_b64 per MFMA: lane l reads component 4q + (l >> 4) of row l & 15; record stride DP + 2 doubles puts the 16 rows
// on distinct bank quadruples), B operand w_j in registers for all row tiles of the chunk.  The matrix pipe then carries
// the D multiply-adds per element and the VALU only the degree-K polynomial and the weighting of the 4 results per lane,
// the two pipes running side by side.  A wave owns 16-column blocks cb = wave, wave + 16, ...; lanes (l & 15) = column,
// (l >> 4) + 4r = row inside the tile (C/D layout of the f64 MFMA); summation order is fixed (bitwise reproducible).
#pragma once
#include "rollout_kernel.h"

namespace gpmpc_hip {

// Gaussian elimination with partial pivoting on [A | RHS] in LDS (row stride ld) by ONE wavefront: same algorithm and pivot
// rule as gauss_solve (first largest |entry| of the column), the D - 1 - k by D + nrhs - 1 - k trailing elements of a step
// spread over the 64 lanes, the back substitution one right-hand side per lane.  A single thread doing this on a 16 x 32
// block took ~0.7 M cycles (dependent LDS round trips) -- 152 of them were 100 ms of a config-5 horizon step.
// Returns det(A) on every lane.  Call from one whole wavefront; ends with the LDS writes visible to that wavefront.
__device__ inline double wave_gauss_solve(double* aug, int D, int nrhs, int ld, int lane) {
    double det = 1.0;
    const int nc = D + nrhs;
    for (int k = 0; k < D; ++k) {
        double best = (lane >= k && lane < D) ? fabs(aug[lane * ld + k]) : -1.0;
        int piv = lane;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double ob = __shfl_xor(best, off, 64);
            const int op = __shfl_xor(piv, off, 64);
            if (ob > best || (ob == best && op < piv)) { best = ob; piv = op; }
        }
        piv = __builtin_amdgcn_readfirstlane(piv);
        if (piv != k) {
            for (int c = lane; c < nc; c += kWave) {
                const double t = aug[k * ld + c];
                aug[k * ld + c] = aug[piv * ld + c];
                aug[piv * ld + c] = t;
            }
            det = -det;
            wave_lds_sync();
        }
        const double pv = aug[k * ld + k];
        det *= pv;
        const double ip = 1.0 / pv;
        const int ncols = nc - 1 - k;
        const int total = (D - 1 - k) * ncols;
        for (int idx = lane; idx < total; idx += kWave) {
            const int rr = idx / ncols;
            const int r = k + 1 + rr, c = k + 1 + (idx - rr * ncols);
            const double f = aug[r * ld + k] * ip;
            aug[r * ld + c] -= f * aug[k * ld + c];
        }
        wave_lds_sync();
    }
    if (lane < nrhs) {
        const int c = D + lane;
        for (int k = D - 1; k >= 0; --k) {
            double sacc = aug[k * ld + c];
            for (int r = k + 1; r < D; ++r) sacc -= aug[k * ld + r] * aug[r * ld + c];
            aug[k * ld + c] = sacc / aug[k * ld + k];
        }
    }
    wave_lds_sync();
    return det;
}

typedef double mfma_d4 __attribute__((ext_vector_type(4)));

// Row stride of the staged records: DP + 2 doubles.  Round-4 experiments on the A-operand reads of the matrix-core pass (lane l:
// component 4 q + (l >> 4) of row l & 15), which the compiler emits as ds_read2_b64 pairs -- serviced per 16-lane group with
// banks taken mod 32, so that at a stride of 18 doubles (36 dwords = 4 mod 32) rows r and r + 8 collide: 16 LDS cycles per
// instruction where 8 suffice, the whole of the 3.44e10 bank-conflict cycles of an N = 4096 launch (= 4 reads x 8 cycles x
// 1.07e9 tile pairs, profiles/r04_c5_late_pmc_t0.txt), 31 % of the kernel's LDS cycles:
//   * an ODD stride (19 doubles: the 16 rows of a group on 16 distinct even banks) removes them -- SQ_LDS_BANK_CONFLICT 3.44e10 -> ~0
//     at the initial state, 6.25e10 -> 2.81e10 (the table gathers of the mid-range exponential) at a late-horizon state
//     (profiles/r04d_c5_odd_stride_pmc.txt) -- and the horizon step takes 481.8 / 580.9 ms against 476.6 / 573.2 (+1 %);
//   * separate ds_read_b64 (half the LDS cycles again, profiles/r04_lds_read_forms.txt) cost an address instruction each on
//     the issue port the fp64 vector and matrix instructions share: 500.3 / 604.9 ms (+5 ... 7 %, profiles/r04b_ab_c5_late_*).
// The LDS conflicts were never on the critical path: the kernel is bound by fp64 issue (matrix + vector instructions keep the
// shared pipe busy for 0.36 s of a 0.48 s step at the initial state, 0.45 of 0.57 s late in the horizon).  Stride kept at 18.
__host__ __device__ constexpr int stream_row_stride(int DP) { return DP + 2; }

// c tiles of TWO 16-row tiles (rows 16 t0 .., 16 t0 + 16 ..) of the stage against the wave's 16 columns.
template <int DP>
__device__ inline void mfma_c_tiles(const double* st, int t0, const double (&bw)[DP / 4], int lane, mfma_d4& c0, mfma_d4& c1) {
    constexpr int RS = stream_row_stride(DP);
    const double* a0p = st + (size_t)(16 * t0 + (lane & 15)) * RS + 2 + (lane >> 4);
    const double* a1p = a0p + 16 * RS;
    double a0[DP / 4], a1[DP / 4];
#pragma unroll
    for (int q = 0; q < DP / 4; ++q) { a0[q] = a0p[4 * q]; a1[q] = a1p[4 * q]; }
    c0 = {0.0, 0.0, 0.0, 0.0};
    c1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < DP / 4; ++q) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[q], bw[q], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[q], bw[q], c1, 0, 0, 0);
    }
}

// Taylor form of one (row chunk, 16-column block) item: ntiles 16-row tiles (processed two at a time; a tile past `ntiles`
// only meets zero weights: zero padding records or the zero lower triangle of T).
//   diagonal pair : sum_i T[i][j] ea_i P_K(c_ij)       off-diagonal : sum_i ra_i P_K(c_ij)
// Tp = &T_a[chunk row 0 + (lane >> 4)][j] for this lane's column.
// The 8 polynomials of an iteration are evaluated as 8 INTERLEAVED Horner chains and summed into 4 accumulators: a dependent
// v_fma_f64 issues only every ~40 cycles, so 8 serial chains of K + 1 links (what the compiler makes of the obvious loop,
// to save registers) cost 8 (K + 1) x 40 cycles per iteration against 512 cycles of matrix-core work.
template <int DP, int K, bool DIAG>
__device__ inline double block_mfma_taylor(const double* st, int ntiles, const double (&bw)[DP / 4], const double* Tp, int N, int lane) {
    constexpr int RS = stream_row_stride(DP);
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const double* wrow = st + (size_t)(lane >> 4) * RS;
    for (int t = 0; t < ntiles; t += 2) {
        double wt[8];
        if (DIAG) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                wt[r] = Tp[(size_t)(16 * t + 4 * r) * N];
                wt[4 + r] = Tp[(size_t)(16 * t + 16 + 4 * r) * N];
            }
        }
        mfma_d4 c0, c1;
        mfma_c_tiles<DP>(st, t, bw, lane, c0, c1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double* w0 = wrow + (size_t)(16 * t + 4 * r) * RS;
            const double* w1 = w0 + 16 * RS;
            if (DIAG) { wt[r] *= w0[0]; wt[4 + r] *= w1[0]; } else { wt[r] = w0[1]; wt[4 + r] = w1[1]; }
        }
        double cv[8], pv[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { cv[r] = c0[r]; cv[4 + r] = c1[r]; }
#pragma unroll
        for (int e = 0; e < 8; ++e) pv[e] = kInvFact[K];
#pragma unroll
        for (int k = K - 1; k >= 0; --k) {
#pragma unroll
            for (int e = 0; e < 8; ++e) pv[e] = fma(pv[e], cv[e], kInvFact[k]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e & 3] = fma(pv[e], wt[e], acc[e & 3]);
    }
    return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

// Mid-range form of the same item (|g.w| beyond the Taylor range, up to kTableMaxArg): exp(ka' + kb' + c) =
// e^ka' e^kb' e^c with the per-point factors already in the row record / column factor (exactly as in the Taylor form),
// and e^c = T[n] P_5(r), n = round(128 c), r = c - n / 128 (exact), |r| <= 1/256, T[n] = exp(n / 128) tabulated in LDS
// (2049 entries), truncation r^6 / 720 <= 5e-18: 11 VALU instructions per element where the general fast_exp of the
// direct form needs 16 plus the two additions that form its argument -- on this part the fp64 vector and matrix
// instructions share one pipe, so every instruction saved per element is time.  Config 5 spends most of its horizon
// here (predicted variances ~1e-2 give |g.w| ~ 0.5 .. 4).
constexpr double kTableMaxArg = 7.99;
constexpr int kTableHalf = 1024;                 // T[kTableHalf + n] = exp(n / 128), |n| <= 1024 (16 KB of LDS)

__device__ inline double table_exp(double c, const double* tab /* centre of the table */) {
    // |r| <= 1 / 256: the degree-5 polynomial truncates at r^6 / 720 <= 5e-18 (spacing 1 / 64 needed degree 6)
    const double n = __builtin_rint(c * 128.0);
    const double r = fma(n, -0.0078125, c);
    const double t = tab[(int)n];
    double q = fma(r, 1.0 / 120, 1.0 / 24);
    q = fma(q, r, 1.0 / 6);
    q = fma(q, r, 0.5);
    q = fma(q, r, 1.0);
    q = fma(q, r, 1.0);
    return t * q;
}

template <int DP, bool DIAG>
__device__ inline double block_mfma_table(const double* st, int ntiles, const double (&bw)[DP / 4], const double* Tp, int N, int lane,
                                          const double* tab) {
    constexpr int RS = stream_row_stride(DP);
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const double* wrow = st + (size_t)(lane >> 4) * RS;
    for (int t = 0; t < ntiles; t += 2) {
        double wt[8];
        if (DIAG) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                wt[r] = Tp[(size_t)(16 * t + 4 * r) * N];
                wt[4 + r] = Tp[(size_t)(16 * t + 16 + 4 * r) * N];
            }
        }
        mfma_d4 c0, c1;
        mfma_c_tiles<DP>(st, t, bw, lane, c0, c1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double* w0 = wrow + (size_t)(16 * t + 4 * r) * RS;
            const double* w1 = w0 + 16 * RS;
            if (DIAG) { wt[r] *= w0[0]; wt[4 + r] *= w1[0]; } else { wt[r] = w0[1]; wt[4 + r] = w1[1]; }
        }
        // the 8 evaluations of table_exp stage by stage: written as 8 calls the compiler emits 8 serial chains of 11 dependent
        // instructions (a dependent fp64 instruction issues every ~40 cycles: 16 cycles per instruction and SIMD even with
        // four wavefronts), interleaved every instruction has 7 independent ones behind it
        double cv[8], nv[8], tv[8], qv[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { cv[r] = c0[r]; cv[4 + r] = c1[r]; }
        // range reduction without rint / convert: adding 1.5 * 2^45 (ulp 2^-7) rounds c to the nearest n / 128 and leaves the
        // integer n in the low mantissa bits (two's complement); n / 128 comes back exactly by subtracting the constant
        constexpr double kShift = 52776558133248.0;                                     // 1.5 * 2^45
#pragma unroll
        for (int e = 0; e < 8; ++e) nv[e] = cv[e] + kShift;
#pragma unroll
        for (int e = 0; e < 8; ++e) tv[e] = tab[(int)__double2loint(nv[e])];
#pragma unroll
        for (int e = 0; e < 8; ++e) cv[e] -= nv[e] - kShift;                            // r = c - n / 128 (exact)
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
        for (int e = 0; e < 8; ++e) acc[e & 3] = fma(qv[e], wt[e], acc[e & 3]);
    }
    return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

// Direct form exp(ka'_i + kb'_j + c_ij) of the same item (record [0] = ka'_i, [1] = beta_ai).
template <int DP, bool DIAG>
__device__ inline double block_mfma_exp(const double* st, int ntiles, const double (&bw)[DP / 4], double kbj, const double* Tp, int N,
                                        int lane, const double* tab) {
    constexpr int RS = stream_row_stride(DP);
    double acc[2] = {0.0, 0.0};
    const double* wrow = st + (size_t)(lane >> 4) * RS;
    for (int t = 0; t < ntiles; t += 2) {
        double wt[8];
        if (DIAG) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                wt[r] = Tp[(size_t)(16 * t + 4 * r) * N];
                wt[4 + r] = Tp[(size_t)(16 * t + 16 + 4 * r) * N];
            }
        }
        mfma_d4 c0, c1;
        mfma_c_tiles<DP>(st, t, bw, lane, c0, c1);
        double arg[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double* w0 = wrow + (size_t)(16 * t + 4 * r) * RS;
            const double* w1 = w0 + 16 * RS;
            arg[r] = w0[0] + kbj + c0[r];
            arg[4 + r] = w1[0] + kbj + c1[r];
            if (!DIAG) { wt[r] = w0[1]; wt[4 + r] = w1[1]; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e & 1] = fma(fast_exp(arg[e], tab), wt[e], acc[e & 1]);
    }
    return acc[0] + acc[1];
}

struct StreamLayout {
    int mu, Sig, m, M, cc, s1, Vs, Sp, aug, red, kb, stage, ints;
    int c_ils2, c_logvar, c_var, c_xr, c_act, c_exptab, c_etab, ush;
    int total;     // doubles
};

__host__ __device__ inline StreamLayout make_stream_layout(int N, int D, int A, int E, int DP, int CH, int HA) {
    StreamLayout L;
    const int P = D * (D + 1) / 2;
    int o = 0;
    L.mu = o;       o += rnd2(D);
    L.Sig = o;      o += 2 * rnd2(D * D);
    L.m = o;        o += rnd2(E);
    L.M = o;        o += rnd2(D);
    L.cc = o;       o += rnd2(D);
    L.s1 = o;       o += rnd2(D * (D + 1));
    L.Vs = o;       o += rnd2(D * D);
    L.Sp = o;       o += rnd2(P);
    L.aug = o;      o += 2 * D * D;
    L.red = o;      o += rnd2(16 * (DP + 1));
    L.kb = o;       o += rnd2(N);
    L.stage = o;    o += rnd2(2 * (CH + 16) * stream_row_stride(DP));      // + 16 rows: the matrix-core items take row tiles two at a time
    L.ints = o;     o += 4;
    L.c_ils2 = o;   o += rnd2(D * E);
    L.c_logvar = o; o += rnd2(D);
    L.c_var = o;    o += rnd2(D);
    L.c_xr = o;     o += rnd2(2 * E);
    L.c_act = o;    o += rnd2(HA);
    L.c_exptab = o; o += 64;
    L.c_etab = o;   o += (DP % 4 == 0 && DP >= 8) ? 2 * kTableHalf + 2 : 0;     // exp(n / 128), |n| <= 1024 (matrix-core pair pass only)
    L.ush = o;      o += (DP == 16) ? 16 * 64 : 0;                                  // per-wavefront hand-off slot of the D = 16 stage fill
    L.total = o;
    return L;
}

template <int DP, int NT>
__global__ __launch_bounds__(NT) void rollout_stream_kernel(const RolloutArgs p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int NW = NT / kWave;
    constexpr int RS = stream_row_stride(DP);
    constexpr int DPC = DP <= 2 ? 2 : (DP <= 4 ? 4 : (DP <= 8 ? 8 : 16));     // lanes that share one row record
    static_assert(NT == 1024, "stage fill maps 64 rows x 16 components onto 1024 threads");
    constexpr bool kMfma = (DP % 4 == 0) && DP >= 8;          // pairwise pass on the fp64 matrix cores
    const int SSTR = (p.CH + 16) * RS;                        // doubles per stage buffer
    const int tid0 = threadIdx.x;
    const int c = blockIdx.x;
    const int D = p.D, N = p.N, A = p.A, E = p.E, H = p.H, CH = p.CH;
    const int DA = D + A;
    const int LD = 2 * D;
    const int SD2 = rnd2(D * D);
    const int RC = (N + CH - 1) / CH;
    const int NCB = (N + 63) / 64;

    const StreamLayout L = make_stream_layout(N, D, A, E, DP, CH, H * A);
    double* s_mu = smem + L.mu;
    double* s_Sig2 = smem + L.Sig;
    double* s_m = smem + L.m;
    double* s_M = smem + L.M;
    double* s_cc = smem + L.cc;
    double* s_s1 = smem + L.s1;
    double* s_Vs = smem + L.Vs;
    double* s_Sp = smem + L.Sp;
    double* s_aug = smem + L.aug;
    double* s_red = smem + L.red;
    double* s_kb = smem + L.kb;
    double* s_stage = smem + L.stage;
    int* s_int = reinterpret_cast<int*>(smem + L.ints);      // [0] Taylor degree of the current pair
    double* s_rdet = smem + L.ints + 2;
    double* c_ils2 = smem + L.c_ils2;
    double* c_logvar = smem + L.c_logvar;
    double* c_var = smem + L.c_var;
    double* c_xr = smem + L.c_xr;
    double* c_act = smem + L.c_act;
    double* c_exptab = smem + L.c_exptab;
    const double* act = p.actions + (size_t)c * H * A;

    for (int i = tid0; i < D; i += NT) { s_mu[i] = p.mu0[i]; c_logvar[i] = p.logvar[i]; c_var[i] = p.var[i]; }
    for (int i = tid0; i < D * D; i += NT) s_Sig2[i] = p.S0[i];
    for (int i = tid0; i < D * E; i += NT) c_ils2[i] = p.ils2[i];
    for (int i = tid0; i < 2 * E; i += NT) c_xr[i] = p.xrange[i];
    for (int i = tid0; i < H * A; i += NT) c_act[i] = act[i];
    for (int i = tid0; i < 64; i += NT) c_exptab[i] = kExp2Tab[i];
    double* c_etab = smem + L.c_etab + kTableHalf;                     // centre of the table
    double* s_ush = smem + L.ush;
    if constexpr (kMfma) {
        for (int i = tid0; i <= 2 * kTableHalf; i += NT) c_etab[i - kTableHalf] = exp((double)(i - kTableHalf) * 0.0078125);
    }
    __syncthreads();
    for (int i = tid0; i < D; i += NT) p.mu_out[((size_t)c * (H + 1)) * D + i] = s_mu[i];
    for (int i = tid0; i < D * D; i += NT) p.Sig_out[((size_t)c * (H + 1)) * D * D + i] = s_Sig2[i];

    int cur = 0;
    for (int t = 0; t < H; ++t) {
        int tid_opaque = tid0;
        asm volatile("" : "+v"(tid_opaque));          // see rollout_kernel.h: stops invariant hoisting
        const int tid = tid_opaque;
        const int lane = tid & 63;
        const int wave = tid >> 6;
        const double* s_Sig = s_Sig2 + cur * SD2;
        double* s_SigNext = s_Sig2 + (cur ^ 1) * SD2;

        for (int i = tid; i < E; i += NT) {
            double v;
            if (i < D) v = s_mu[i];
            else if (i < DA) v = c_act[t * A + (i - D)];
            else v = p.time0 + (double)t;
            s_m[i] = v;
        }
        __syncthreads();

        // ---- mean part: M_a, V_a for one output dimension at a time (gp_model.py:140-153) -------------------
        for (int a = 0; a < D; ++a) {
            if (wave == 0) {
                double prodil = 1.0;
                for (int i = 0; i < D; ++i) prodil *= c_ils2[a * E + i];
                for (int idx = lane; idx < D * D; idx += kWave) {
                    const int i = idx / D, j = idx - i * D;
                    s_aug[i * LD + j] = s_Sig[idx] + (i == j ? 1.0 / c_ils2[a * E + i] : 0.0);
                    s_aug[i * LD + D + j] = (i == j ? 1.0 : 0.0);
                }
                wave_lds_sync();
                const double detA = wave_gauss_solve(s_aug, D, D, LD, lane);
                if (lane == 0) s_cc[a] = c_var[a] / sqrt(detA * prodil);
            }
            __syncthreads();
            double acc[DP + 1];
#pragma unroll
            for (int k = 0; k <= DP; ++k) acc[k] = 0.0;
            const double* Ai = s_aug + D;
            for (int pt = tid; pt < N; pt += NT) {
                double nu[DP];
#pragma unroll
                for (int d = 0; d < DP; ++d) nu[d] = (d < D) ? (p.Xt[(size_t)d * N + pt] - s_m[d]) : 0.0;
                double q = 0.0;
#pragma unroll
                for (int i = 0; i < DP; ++i) {
                    if (i < D) {
                        double r = 0.0;
#pragma unroll
                        for (int j = 0; j < DP; ++j)
                            if (j < D) r = fma(Ai[i * LD + j], nu[j], r);
                        q = fma(nu[i], r, q);
                    }
                }
                for (int e = D; e < E; ++e) {
                    const double v = p.Xt[(size_t)e * N + pt] - s_m[e];
                    q = fma(v * v, c_ils2[a * E + e], q);
                }
                const double lb = exp(-0.5 * q) * p.beta[(size_t)a * N + pt];
                acc[0] += lb;
#pragma unroll
                for (int d = 0; d < DP; ++d) acc[1 + d] = fma(lb, nu[d], acc[1 + d]);
            }
#pragma unroll
            for (int k = 0; k <= DP; ++k) {
                const double v = wave_sum(acc[k]);
                if (lane == 0) s_red[wave * (DP + 1) + k] = v;
            }
            __syncthreads();
            if (tid <= D) {
                double s = 0.0;
                for (int w = 0; w < NW; ++w) s += s_red[w * (DP + 1) + tid];
                s_s1[a * (D + 1) + tid] = s;
            }
            __syncthreads();
            if (tid < D) {
                double s = 0.0;
                for (int j = 0; j < D; ++j) s = fma(Ai[tid * LD + j], s_s1[a * (D + 1) + 1 + j], s);
                s_Vs[tid * D + a] = s_cc[a] * s;                        // state rows of V (:153)
                if (tid == 0) s_M[a] = s_cc[a] * s_s1[a * (D + 1)];     // M_a (:152)
            }
            __syncthreads();
        }

        // ---- covariance part: one output pair at a time (gp_model.py:156-178) ------------------------------
        int q = 0;
        for (int a = 0; a < D; ++a) {
            for (int b = a; b < D; ++b, ++q) {
                const bool diag = (a == b);
                if (wave == 0) {
                    for (int idx = lane; idx < D * D; idx += kWave) {
                        const int i = idx / D, j = idx - i * D;
                        const double dab = c_ils2[a * E + j] + c_ils2[b * E + j];
                        s_aug[i * LD + j] = s_Sig[idx] * dab + (i == j ? 1.0 : 0.0);
                        s_aug[i * LD + D + j] = s_Sig[idx];
                    }
                    wave_lds_sync();
                    const double detR = wave_gauss_solve(s_aug, D, D, LD, lane);
                    const double* Z = s_aug + D;
                    // |g_i . w_j| <= cmax = sum_dd' |Z_dd'| umax_d wmax_d' over the data range of the memory points
                    double cpart = 0.0;
                    for (int idx = lane; idx < D * D; idx += kWave) {
                        const int i = idx / D, j = idx - i * D;
                        const double mi = s_mu[i], mj = s_mu[j];
                        const double ui = fmax(fabs(c_xr[i] - mi), fabs(c_xr[E + i] - mi)) * c_ils2[a * E + i];
                        const double wj = fmax(fabs(c_xr[j] - mj), fabs(c_xr[E + j] - mj)) * c_ils2[b * E + j];
                        cpart = fma(fabs(Z[i * LD + j]) * ui, wj, cpart);
                    }
                    const double cmax = wave_sum(cpart);
                    if (lane == 0) {
                        s_rdet[0] = 1.0 / sqrt(detR);
                        int K = 0;                                       // 0: direct exp; 1..14: Taylor degree; 15: table form
                        if (p.force_path != 1) {
                            if (cmax <= kTaylorMaxArg[kMaxTaylor] && p.force_path != 4) {
                                K = 1;
                                for (int k = 1; k < kMaxTaylor; ++k) K += (cmax > kTaylorMaxArg[k]) ? 1 : 0;
                            } else if (kMfma && cmax <= kTableMaxArg) K = kMaxTaylor + 1;
                        }
                        s_int[0] = K;
                    }
                }
                __syncthreads();
                const int K = __builtin_amdgcn_readfirstlane(s_int[0]);
                const double* Z = s_aug + D;

                // column factors: exp(kb'_j) beta_bj  (Taylor form), kb'_j (direct form); diagonal pair: ea_j
                for (int j = tid; j < N; j += NT) {
                    double w[DP];
                    double ks = 0.0;
                    double xv[DP];
#pragma unroll
                    for (int d = 0; d < DP; ++d) xv[d] = p.Xt[(size_t)(d < D ? d : D - 1) * N + j];      // independent loads
                    double xe[8];
                    const int ne = E - D;                                                               // <= 8 action / time inputs
#pragma unroll
                    for (int e = 0; e < 8; ++e) xe[e] = p.Xt[(size_t)(e < ne ? D + e : D) * N + j];
#pragma unroll
                    for (int d = 0; d < DP; ++d) {
                        const double nu = (d < D) ? (xv[d] - s_m[d]) : 0.0;
                        w[d] = (d < D) ? nu * c_ils2[b * E + d] : 0.0;
                        ks = fma(nu, w[d], ks);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (e < ne) {
                            const double v = xe[e] - s_m[D + e];
                            ks = fma(v * v, c_ils2[b * E + D + e], ks);
                        }
                    }
                    double qb = 0.0;
#pragma unroll
                    for (int i = 0; i < DP; ++i) {
                        if (i < D) {
                            double zw = 0.0;
#pragma unroll
                            for (int jj = 0; jj < DP; ++jj)
                                if (jj < D) zw = fma(Z[i * LD + jj], w[jj], zw);
                            qb = fma(w[i], zw, qb);
                        }
                    }
                    const double kb = c_logvar[b] - 0.5 * ks + 0.5 * qb;
                    s_kb[j] = (K > 0) ? (diag ? exp(kb) : exp(kb) * p.beta[(size_t)b * N + j]) : kb;
                }

                // row records of one chunk -> LDS stage; DPC lanes share a row (lane group = one record)
                // D = 16: the 16 lanes of a row share the work -- lane `comp` loads ONE state coordinate (the old form had
                // every lane load all 20 and spend ~70 instructions on their addresses), u goes through a 64-double LDS
                // slot per wavefront, g_comp = sum_d Z[d][comp] u_d, and ka' is one DPP row reduction of the per-lane
                // pieces -1/2 nu u + 1/2 u g
                auto fill_stage16 = [&](int r, double* stage) {
                    const int comp = tid & 15;
                    double* ush = s_ush + wave * 64;
                    for (int trow = tid >> 4; trow < CH; trow += NT / 16) {
                        const int i = r * CH + trow;
                        const bool in = i < N;
                        const int ic = in ? i : N - 1;
                        double u = 0.0, piece = 0.0;
                        if (comp < D) {
                            const double nu = p.Xt[(size_t)comp * N + ic] - s_m[comp];
                            u = nu * c_ils2[a * E + comp];
                            piece = -0.5 * nu * u;
                        }
                        if (comp < E - D) {
                            const double v = p.Xt[(size_t)(D + comp) * N + ic] - s_m[D + comp];
                            piece = fma(-0.5 * v * v, c_ils2[a * E + D + comp], piece);
                        }
                        ush[lane] = u;
                        wave_lds_sync();
                        const double* ug = ush + (lane & 48);
                        double gcomp = 0.0;
                        if (comp < D) {
#pragma unroll
                            for (int d = 0; d < 16; ++d)
                                if (d < D) gcomp = fma(Z[d * LD + comp], ug[d], gcomp);                  // g = Z^T u
                        }
                        wave_lds_sync();                                                               // slot free for the next row group
                        piece = fma(0.5 * u, gcomp, piece);
                        piece += dpp_shifted<0x111, 0xf>(piece);                                       // row_shr 1, 2, 4, 8: lane 15 of the
                        piece += dpp_shifted<0x112, 0xf>(piece);                                       // row holds the sum of its 16 lanes
                        piece += dpp_shifted<0x114, 0xf>(piece);
                        piece += dpp_shifted<0x118, 0xf>(piece);
                        double* rec = stage + (size_t)trow * RS;
                        rec[2 + comp] = in ? gcomp : 0.0;
                        if (comp == 15) {
                            double r0 = 0.0, r1 = 0.0;
                            if (in) {
                                const double ka = c_logvar[a] + piece;
                                const double ba = p.beta[(size_t)a * N + i];
                                if (K > 0) { r0 = exp(ka); r1 = r0 * ba; } else { r0 = ka; r1 = ba; }
                            }
                            rec[0] = r0;
                            rec[1] = r1;
                        }
                    }
                };
                auto fill_stage_gen = [&](int r, double* stage) {
                  const int comp = tid % DPC;
                  for (int trow = tid / DPC; trow < CH; trow += NT / DPC) {
                    {
                        const int i = r * CH + trow;
                        double gcomp = 0.0, part = 0.0, ks = 0.0;
                        if (i < N) {
                            double xv[DP];
#pragma unroll
                            for (int d = 0; d < DP; ++d) xv[d] = p.Xt[(size_t)(d < D ? d : D - 1) * N + i];      // independent loads
                            double xe[8];
                            const int ne = E - D;
#pragma unroll
                            for (int e = 0; e < 8; ++e) xe[e] = p.Xt[(size_t)(e < ne ? D + e : D) * N + i];
#pragma unroll
                            for (int d = 0; d < DP; ++d) {
                                if (d < D) {
                                    const double nu = xv[d] - s_m[d];
                                    const double u = nu * c_ils2[a * E + d];
                                    ks = fma(nu, u, ks);
                                    if (comp < D) gcomp = fma(Z[d * LD + comp], u, gcomp);      // g = Z^T u
                                    if (d == comp) part = u;
                                }
                            }
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                if (e < ne) {
                                    const double v = xe[e] - s_m[D + e];
                                    ks = fma(v * v, c_ils2[a * E + D + e], ks);
                                }
                            }
                            part *= gcomp;                                                   // u_comp g_comp
                        }
                        double qa = part;                                                    // u^T Z u over the DPC lanes
#pragma unroll
                        for (int off = 1; off < DPC; off <<= 1) qa += __shfl_xor(qa, off, 64);
                        double* rec = stage + (size_t)trow * RS;
                        if (comp < DP) rec[2 + comp] = gcomp;
                        if (comp == 0) {
                            double r0 = 0.0, r1 = 0.0;
                            if (i < N) {
                                const double ka = c_logvar[a] - 0.5 * ks + 0.5 * qa;
                                const double ba = p.beta[(size_t)a * N + i];
                                if (K > 0) { r0 = exp(ka); r1 = r0 * ba; } else { r0 = ka; r1 = ba; }
                            }
                            rec[0] = r0;
                            rec[1] = r1;
                        }
                    }
                  }
                };
                auto fill_stage = [&](int r, double* stage) {
                    if constexpr (DP == 16) fill_stage16(r, stage); else fill_stage_gen(r, stage);
                };
                fill_stage(0, s_stage);
                __syncthreads();

                double acc = 0.0;
                for (int r = 0; r < RC; ++r) {
                    if (r + 1 < RC) fill_stage(r + 1, s_stage + ((r + 1) & 1) * SSTR);
                    const double* rec = s_stage + (r & 1) * SSTR;
                    int nch = N - r * CH;
                    if (nch > CH) nch = CH;
                    if constexpr (kMfma) {
                        const int tiles_in_chunk = (nch + 15) >> 4;             // rows past the data are zero records
                        const int NCB16 = (N + 15) >> 4;
                        // column operands of an item (B operand w_j, column factor): loaded one item ahead, so that their L2
                        // latency is covered by the previous item's matrix-core work
                        auto col_operands = [&](int cb, double (&bwv)[DP / 4], double& kbv) {
                            int j = cb * 16 + (lane & 15);
                            if (j >= N) j = N - 1;
#pragma unroll
                            for (int qq = 0; qq < DP / 4; ++qq) {
                                const int k = 4 * qq + (lane >> 4);
                                const double x = p.Xt[(size_t)(k < D ? k : D - 1) * N + j];
                                bwv[qq] = (k < D) ? (x - s_m[k < D ? k : 0]) * c_ils2[b * E + (k < D ? k : 0)] : 0.0;
                            }
                            kbv = s_kb[j];
                        };
                        int cb = wave;
                        if (diag) { const int first = (r * CH) >> 4; cb += ((first - wave + NW - 1) / NW) * NW; if (cb < wave) cb = wave; }
                        double bw[DP / 4], kbj;
                        if (cb < NCB16) col_operands(cb, bw, kbj);
                        for (; cb < NCB16; cb += NW) {
                            double bwn[DP / 4], kbn;
                            const int cbn = (cb + NW < NCB16) ? cb + NW : cb;
                            col_operands(cbn, bwn, kbn);
                            const int j0 = cb * 16;
                            int ntiles = tiles_in_chunk;
                            if (diag) { const int lim = (j0 + 16 - r * CH + 15) >> 4; if (lim < ntiles) ntiles = lim; }
                            const int j = j0 + (lane & 15);
                            const bool valid = j < N;
                            const int jc = valid ? j : N - 1;
                            const double* Tp = p.Tm + ((size_t)a * (N + kTPad) + (size_t)r * CH + (lane >> 4)) * N + jc;
                            double v;
                            if (K == 0) {
                                v = diag ? block_mfma_exp<DP, true>(rec, ntiles, bw, kbj, Tp, N, lane, c_exptab)
                                         : block_mfma_exp<DP, false>(rec, ntiles, bw, kbj, Tp, N, lane, c_exptab);
                                v *= diag ? 2.0 : p.beta[(size_t)b * N + jc];
                            } else {
#define GPMPC_TAYLOR_BLOCK(KK) (diag ? block_mfma_taylor<DP, KK, true>(rec, ntiles, bw, Tp, N, lane) \
                                     : block_mfma_taylor<DP, KK, false>(rec, ntiles, bw, Tp, N, lane))
                                if (K <= 2) v = GPMPC_TAYLOR_BLOCK(2);
                                else if (K <= 4) v = GPMPC_TAYLOR_BLOCK(4);
                                else if (K <= 6) v = GPMPC_TAYLOR_BLOCK(6);
                                else if (K <= 8) v = GPMPC_TAYLOR_BLOCK(8);
                                else if (K <= 10) v = GPMPC_TAYLOR_BLOCK(10);
                                else if (K <= 12) v = GPMPC_TAYLOR_BLOCK(12);
                                else if (K <= 14) v = GPMPC_TAYLOR_BLOCK(14);
                                else v = diag ? block_mfma_table<DP, true>(rec, ntiles, bw, Tp, N, lane, c_etab)
                                              : block_mfma_table<DP, false>(rec, ntiles, bw, Tp, N, lane, c_etab);
#undef GPMPC_TAYLOR_BLOCK
                                v *= diag ? 2.0 * kbj : kbj;
                            }
                            acc += valid ? v : 0.0;
#pragma unroll
                            for (int qq = 0; qq < DP / 4; ++qq) bw[qq] = bwn[qq];
                            kbj = kbn;
                        }
                        __syncthreads();
                        continue;
                    }
                    nch = (nch + 3) & ~3;                                   // rows past the data are zero records
                    for (int cb = wave; cb < NCB; cb += NW) {
                        const int j0 = cb * 64;
                        if (diag && j0 + 63 < r * CH) continue;             // T is zero below its diagonal
                        int nrows = nch;
                        if (diag) { const int lim = (j0 + 64 - r * CH + 3) & ~3; if (lim < nrows) nrows = lim; }
                        const int j = j0 + lane;
                        const bool valid = j < N;
                        const int jc = valid ? j : N - 1;
                        double w[DP];
#pragma unroll
                        for (int d = 0; d < DP; ++d) w[d] = (d < D) ? (p.Xt[(size_t)d * N + jc] - s_m[d]) * c_ils2[b * E + d] : 0.0;
                        const double kbj = s_kb[jc];
                        const double* Tp = p.Tm + ((size_t)a * (N + kTPad) + (size_t)r * CH) * N + jc;
                        double v;
                        if (K == 0) {
                            v = item_exp<DP>(rec, nrows, w, kbj, diag, Tp, N, c_exptab);
                            v *= diag ? 2.0 : p.beta[(size_t)b * N + jc];
                        } else {
                            if (K <= 2) v = item_taylor<DP, 2>(rec, nrows, w, diag, Tp, N);
                            else if (K <= 4) v = item_taylor<DP, 4>(rec, nrows, w, diag, Tp, N);
                            else if (K == 5) v = item_taylor<DP, 5>(rec, nrows, w, diag, Tp, N);
                            else if (K == 6) v = item_taylor<DP, 6>(rec, nrows, w, diag, Tp, N);
                            else if (K <= 8) v = item_taylor<DP, 8>(rec, nrows, w, diag, Tp, N);
                            else if (K <= 10) v = item_taylor<DP, 10>(rec, nrows, w, diag, Tp, N);
                            else if (K <= 12) v = item_taylor<DP, 12>(rec, nrows, w, diag, Tp, N);
                            else v = item_taylor<DP, 14>(rec, nrows, w, diag, Tp, N);
                            v *= diag ? 2.0 * kbj : kbj;
                        }
                        acc += valid ? v : 0.0;
                    }
                    __syncthreads();
                }
                acc = wave_sum(acc);
                if (lane == 0) s_red[wave] = acc;
                __syncthreads();
                if (tid == 0) {
                    double s = 0.0;
                    for (int w = 0; w < NW; ++w) s += s_red[w];
                    s_Sp[q] = s * s_rdet[0];
                }
                __syncthreads();
            }
        }

        // ---- state update (gp_model.py:105-108, 177-178) -------------------------------------------------
        for (int idx = tid; idx < D * D; idx += NT) {
            const int i = idx / D, j = idx - i * D;
            const int a = i < j ? i : j, b = i < j ? j : i;
            const int qq = a * D - (a * (a - 1)) / 2 + (b - a);
            const double S = s_Sp[qq] - s_M[i] * s_M[j] + (i == j ? c_var[i] : 0.0);
            double cij = 0.0, cji = 0.0;
            for (int k = 0; k < D; ++k) {
                cij = fma(s_Sig[i * D + k], s_Vs[k * D + j], cij);
                cji = fma(s_Sig[j * D + k], s_Vs[k * D + i], cji);
            }
            const double v = S + s_Sig[idx] + (cij + cji);
            s_SigNext[idx] = v;
            p.Sig_out[((size_t)c * (H + 1) + (t + 1)) * D * D + idx] = v;
        }
        for (int i = tid; i < D; i += NT) {
            const double v = s_mu[i] + s_M[i];
            s_mu[i] = v;
            p.mu_out[((size_t)c * (H + 1) + (t + 1)) * D + i] = v;
        }
        cur ^= 1;
        __syncthreads();
    }
}

}  // namespace gpmpc_hip
