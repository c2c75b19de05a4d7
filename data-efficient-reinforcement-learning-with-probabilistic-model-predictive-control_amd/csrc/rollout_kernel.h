// rollout_kernel.h -- the GP-MPC inner loop for gfx950 (MI355X, CDNA4).
//
// One workgroup owns one candidate action sequence for the whole horizon: the H-step
// recurrence (reference rl_gp_mpc/control_objects/models/gp_model.py:60-110) is sequential
// per candidate but candidates are independent, so no inter-workgroup traffic exists and the
// whole batch is ONE launch.  Per horizon step the workgroup evaluates the moment-matched GP
// prediction (gp_model.py:112-180) in a restructured, memory-lean form:
//
//   input covariance is non-zero only in its state block, so every E x E solve of the
//   reference collapses to a D x D one:
//     mean   : q_ai = nu_i^T (Sigma + Lambda_a)^-1 nu_i           (= iN B^-1 iN^T, :140-148)
//     cov    : L_ab,ij = exp(ka'_i + kb'_j + g_i . w_j)            (= exp(k_a + k_b + maha), :161-169)
//              with Z = R_ab^-1 Sigma (= 2Q, :163), u_i = nu_i/l_a^2, w_j = nu_j/l_b^2,
//              ka'_i = k_ai + u_i^T Z u_i / 2, kb'_j = k_bj + w_j^T Z w_j / 2, g_i = Z u_i
//     S_ab   = [sum_ij beta_ai L_ab,ij beta_bj - d_ab sum_ij iK_a,ij L_aa,ij]/sqrt|R_ab| ...   (:170-178)
//   Because |g_i . w_j| <= cmax is bounded per (pair, step) from the data range and Z, the factor
//   exp(g_i . w_j) is evaluated by a Taylor polynomial of the degree K that makes the truncation
//   error < 2^-54 relative (exact to fp64 rounding), the other two factors exp(ka'_i), exp(kb'_j)
//   are per-point; for an off-diagonal pair the double sum then separates over monomials,
//     sum_ij ra_i rb_j exp(g_i . w_j) = sum_|alpha|<=K (sum_i ra_i g_i^alpha)(sum_j rb_j w_j^alpha)/alpha!,
//   i.e. O(N C(D+K,D)) instead of O(N^2) (used for D <= 4); when cmax is too large for K <= 14 the
//   pair falls back to the direct exp(ka' + kb' + g.w) evaluation.
//   The (D,D,N,N) tensors of the reference (:166,169-171) never exist: the N x N pairwise
//   work of each output pair a <= b is streamed through registers; for a == b the two sums
//   are merged through T_a = beta_a beta_a^T - iK_a and only i <= j is visited (L_aa and T_a
//   are symmetric).
//
// Mapping to the hardware: lanes of a 64-wide wavefront own consecutive columns j (coalesced
// T_a reads, conflict-free LDS reads), the row operands (ka', beta_a, g_i) are LDS broadcasts,
// waves pull (pair, row-chunk, 64-column) work items from an LDS counter so that the
// triangular diagonal pairs balance, and every reduction is a fixed-order wavefront butterfly
// followed by a fixed-order sum, so results are bitwise reproducible run to run.
#pragma once
#include "gpmpc_internal.h"
#include <type_traits>

#if defined(GPMPC_PROF_ON)
// phase profile of workgroup 0 (debug build only): cycles between consecutive trace points, summed over steps
#define GPMPC_TRACE(id) do { if (threadIdx.x == 0 && blockIdx.x == 0) { long long now_ = __builtin_readcyclecounter(); \
    prof_acc[id] += now_ - prof_last; prof_last = now_; } \
    if (CL && threadIdx.x == 0 && (blockIdx.x & 7) == 0) { long long w_ = wall_clock64(); prof_wall[id] += w_ - prof_wlast; prof_wlast = w_; } } while (0)
#elif defined(GPMPC_TRACE_ON)
#define GPMPC_TRACE(id) do { if (threadIdx.x == 0 && blockIdx.x == 0) printf("trace %d t=%d\n", id, t_dbg); } while (0)
#else
#define GPMPC_TRACE(id) do {} while (0)
#endif

namespace gpmpc_hip {

// An 8-byte LDS read that stays ONE ds_read_b64.  The compiler pairs neighbouring 8-byte LDS reads of one base register into
// ds_read2_b64, which gfx950 services as two accesses of four 16-lane groups each with banks taken mod 32: 8 LDS cycles for
// 16 bytes per lane even for a broadcast (two ds_read_b64: 2 + 2), and for the A operand of the fp64 matrix instruction (row
// l & 15 of a stage with an 18-double row stride) rows r and r + 8 of a group collide on top (16 cycles against 4); measured:
// profiles/r04_lds_read_forms.txt.  An opaque copy of the address gives the read a base register of its own, which the
// load/store optimiser cannot pair.  (-DGPMPC_LDS_MERGED: the compiler's pairing, kept for the A/B build `make ab`.)
typedef const __attribute__((address_space(3))) double* lds_cptr;
__device__ inline double lds_b64(const double* p) {
#if defined(GPMPC_LDS_MERGED)
    return *p;
#else
    lds_cptr q = (lds_cptr)p;
    asm volatile("" : "+v"(q));
    return *q;
#endif
}

// Row records of the pairwise pass (one per memory point and output pair, broadcast from LDS to every lane of a wavefront).
//   RowRecPacked<DP>  : ea_i | ra_i | g_i (DP), stride DP + 2 -- the staged records of rollout_stream_kernel.h;
//   RowRecAligned<DP> : g_i (DP) | ea_i | ra_i | pad, stride rounded up to an EVEN number of doubles, records 16-byte aligned and
//                       read as ds_read_b128: 4 LDS cycles per 16 bytes where the compiler's pairing of 8-byte reads
//                       (ds_read2_b64) takes 8 -- at D <= 4 the fused-horizon kernel is as close to the LDS return rate as to
//                       the fp64 issue rate, and a diagonal pair needs only the leading DP + 1 doubles (DP = 3: two reads).
typedef double lds_d2 __attribute__((ext_vector_type(2)));
// the leading NL doubles of a 16-byte aligned LDS record as ds_read_b128 (NV >= NL rounded up to even)
template <int NL, int NV>
__device__ inline void lds_load_pairs(const double* rec, double (&v)[NV]) {
    static_assert(NV >= ((NL + 1) & ~1), "destination too short");
    const lds_d2* r2 = reinterpret_cast<const lds_d2*>(rec);
#pragma unroll
    for (int k = 0; k < (NL + 1) / 2; ++k) { const lds_d2 t = r2[k]; v[2 * k] = t.x; v[2 * k + 1] = t.y; }
}

template <int DP>
struct RowRecPacked {
    static constexpr int RS = DP + 2, EA = 0, RA = 1, G = 2;
    // the leading `NL` doubles of a record into v
    template <int NL>
    static __device__ inline void load(const double* rec, double (&v)[RS]) {
#pragma unroll
        for (int k = 0; k < NL; ++k) v[k] = rec[k];
    }
};
template <int DP>
struct RowRecAligned {
    static constexpr int RS = (DP + 3) & ~1, EA = DP, RA = DP + 1, G = 0;
    template <int NL>
    static __device__ inline void load(const double* rec, double (&v)[RS]) { lds_load_pairs<NL>(rec, v); }
};

constexpr int kMaxTaylor = 14;     // highest Taylor degree of exp(g.w); beyond that: direct exp path
constexpr int kMaxMono = 256;      // most monomials of the separable (off-diagonal) evaluation

// ------------------------------------------------------------------------------------------
// LDS / scratch layout (offsets in doubles), shared by host (sizing) and device (carving).
constexpr int kLaneMapSlots = 32;  // most work-item slots of a diagonal pair whose lane assignment is tabulated in LDS
constexpr int kLaneMapSlotsCluster = 192;      // ... in the cooperative form, whose row chunks are short (many slots per pair)
__host__ __device__ inline int tri_ints(int N) { return (N + 7) / 8 + 2; }      // s_tri: row chunks of >= 8 rows, + 1, + 1 spare

struct Layout {
    int mu, Sig, m, M, cc, s1, Vs, Sp, misc, rdet, aug, part, mom, ints, lmap, cl;
    int c_ils2, c_logvar, c_var, c_xr, c_act, c_exptab, c_monow, c_monoe, c_X;    // read-only tables copied to LDS once
    int lds_total;     // doubles of LDS
    // per-point arrays (LDS)
    int nu, lb, rows, kb;
    int pp_total;      // doubles of the per-point block
};

__host__ __device__ inline int rnd2(int x) { return (x + 1) & ~1; }

constexpr int kClusterScratch = 16 * 64 / 8;     // the cooperative form's per-wavefront problem lists (16 wavefronts x 64 bytes)

__host__ __device__ inline Layout make_layout(int N, int D, int A, int E, int G, int DP, int wpp, int CM, int CH, int HA,
                                              bool x_in_lds, bool cluster = false, int lds_pairs = 0) {
    const int GL = (cluster && lds_pairs > 0) ? lds_pairs : G;      // pairs whose per-point records live in LDS (cooperative form: the member's own)
    Layout L;
    const int P = D * (D + 1) / 2;
    int o = 0;
    L.mu = o;   o += rnd2(D);
    L.Sig = o;  o += 2 * rnd2(D * D);           // double-buffered
    L.m = o;    o += rnd2(E);
    L.M = o;    o += rnd2(D);
    L.cc = o;   o += rnd2(D);
    L.s1 = o;   o += rnd2(D * (D + 1));
    L.Vs = o;   o += rnd2(D * D);
    L.Sp = o;   o += rnd2(P);
    L.misc = o; o += 8;
    L.rdet = o; o += rnd2(G);
    L.aug = o;  o += (D + G) * 2 * D * D;       // D mean problems + G pair problems, [A | RHS]
    L.part = o; o += rnd2(G * wpp);
    L.mom = o;  o += G * 2 * rnd2(CM);
    // pa[P], pb[P], K[G], counter, noff, off[G], mcum[16], tri[RC+1], nslot[G], nitems (ints); then the step's work-item list
    // (16-bit entries: D mean items + at most G * wpp pair items)
    L.ints = o; o += rnd2((2 * P + 3 * G + 8 + 16 + tri_ints(N) + 1) / 2) + rnd2((D + G * wpp + 3) / 4);
    // lane map of the diagonal pairs' work items: (row chunk, column unit) per (slot, lane) + rows per slot, for up to
    // kLaneMapSlots slots (state-independent: filled once per launch)
    const int lms = cluster ? kLaneMapSlotsCluster : kLaneMapSlots;
    L.lmap = o; o += rnd2(((wpp < lms ? wpp : lms) * 65 + 1) / 2);
    // cooperative form: totals of the separable pairs [G] | ordinals of the pairs among the diagonal / off-diagonal ones (ints) [G] |
    // failure flag + spare (ints) [2] | per-wavefront problem lists of the per-point pass
    // | static owner tables (bytes): item slots [G * wpp], separable pairs [G], pairs needed in element-wise form [G], mean sums [16]
    // | this member's items of the step (16-bit codes) [G * wpp + 16] + their number
    // | LDS slot of each pair's records (ints) [G] | element-wise item slots of each pair this member owns (64-bit masks) [G]
    L.cl = o;       o += cluster ? rnd2(G) + rnd2((G + 1) / 2) + 2 + kClusterScratch + rnd2((G * wpp + 2 * G + 16 + 7) / 8) +
                                   rnd2((G * wpp + 16 + 4 + 3) / 4) + rnd2((G + 1) / 2) + rnd2(G) : 0;     // (even: the row records behind are read 16 bytes at a time)
    L.c_ils2 = o;   o += rnd2(D * E);
    L.c_logvar = o; o += rnd2(D);
    L.c_var = o;    o += rnd2(D);
    L.c_xr = o;     o += rnd2(2 * E);
    L.c_act = o;    o += rnd2(HA);
    L.c_exptab = o; o += 64;
    L.c_monow = o;  o += rnd2(CM);
    L.c_monoe = o;  o += rnd2((CM + 1) / 2);           // packed exponents, one int per monomial
    L.c_X = o;      o += x_in_lds ? rnd2(E * N) : 0;   // X^T cached for the per-point pass when it fits
    int q = o;
    L.nu = q;   q += rnd2(D * N);
    L.lb = q;   q += rnd2(D * N);
    L.rows = q; q += GL * (N + CH) * ((DP + 3) & ~1);   // RowRecAligned<DP>::RS; + CH zero rows per pair (lanes of a wave share the trip count)
    L.kb = q;   q += rnd2(GL * N);
    L.lds_total = q;
    L.pp_total = q - o;
    return L;
}

// ------------------------------------------------------------------------------------------
// Wavefront sum on the DPP crossbar (no LDS round trips): inclusive scan inside each row of 16 lanes
// (row_shr 1,2,4,8), then row_bcast:15 / row_bcast:31 carry the row totals up; lane 63 holds the
// total, which is broadcast through an SGPR.  Fixed order => bitwise reproducible.
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_shifted(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}

__device__ inline double wave_sum(double v) {
    v += dpp_shifted<0x111, 0xf>(v);      // row_shr:1
    v += dpp_shifted<0x112, 0xf>(v);      // row_shr:2
    v += dpp_shifted<0x114, 0xf>(v);      // row_shr:4
    v += dpp_shifted<0x118, 0xf>(v);      // row_shr:8
    v += dpp_shifted<0x142, 0xa>(v);      // row_bcast:15 -> rows 1, 3
    v += dpp_shifted<0x143, 0xc>(v);      // row_bcast:31 -> rows 2, 3
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// LDS hand-off between lanes of ONE wavefront (no workgroup barrier): LDS operations of a wave
// complete in issue order; the fences keep the compiler from moving accesses across.
__device__ inline void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// (Measured and not adopted, round 5 -- the serial phases of a horizon step, P1 3.6 k + P4 1.5 k + P5 1.4 k of 33 k cycles at config 2:
//  (i) 1 / x and 1 / sqrt(x) of the small algebra from the hardware seed + Newton steps instead of the IEEE division and
//  sqrt-then-divide: 0.3738 ms per 256 rollouts either way, P1 unchanged -- the phase is not bound by those chains;
//  (ii) "fused tail": the wavefront that finishes a pair's last work item forms the pair's total, the one that forms the step's
//  last total updates the state, three barriers per step instead of five: P4 + P5 2.9 k -> 0.4 k cycles, but the queue phase
//  + 2.3 k and P1 + 0.7 k -- the chain last item -> total -> state update is serial wherever it runs; 0.411 vs 0.409 ms, and the
//  two-workgroups-per-CU regime (B = 4096) lost its second workgroup to the registers.  profiles/r05f_*, r05g_*.)
// Gaussian elimination with partial pivoting on an augmented [A | RHS] block (row stride ld).
// The solution replaces the RHS; returns det(A).  Same algorithm class (LU, partial pivoting)
// as the torch.linalg.solve / torch.det calls of the reference (gp_model.py:146,150,163,176).
__device__ inline double gauss_solve(double* aug, int D, int nrhs, int ld) {
    double det = 1.0;
    const int nc = D + nrhs;
    for (int k = 0; k < D; ++k) {
        int piv = k;
        double best = fabs(aug[k * ld + k]);
        for (int r = k + 1; r < D; ++r) {
            double v = fabs(aug[r * ld + k]);
            if (v > best) { best = v; piv = r; }
        }
        if (piv != k) {
            for (int c = k; c < nc; ++c) {
                double t = aug[k * ld + c];
                aug[k * ld + c] = aug[piv * ld + c];
                aug[piv * ld + c] = t;
            }
            det = -det;
        }
        const double pv = aug[k * ld + k];
        det *= pv;
        const double ip = 1.0 / pv;
        for (int r = k + 1; r < D; ++r) {
            const double f = aug[r * ld + k] * ip;
            for (int c = k + 1; c < nc; ++c) aug[r * ld + c] -= f * aug[k * ld + c];
        }
    }
    for (int k = D - 1; k >= 0; --k) {
        const double ip = 1.0 / aug[k * ld + k];
        for (int c = D; c < nc; ++c) {
            double s = aug[k * ld + c];
            for (int r = k + 1; r < D; ++r) s -= aug[k * ld + r] * aug[r * ld + c];
            aug[k * ld + c] = s * ip;
        }
    }
    return det;
}

// Register variant for small (padded) dimensions: [A | RHS] lives in registers, every loop is
// unrolled, the pivot row is brought up with selects.  Rows/cols >= D are identity padding.
template <int DP>
__device__ inline double gauss_solve_reg(double (&a)[DP][2 * DP]) {
    double det = 1.0;
#pragma unroll
    for (int k = 0; k < DP; ++k) {
        // partial pivoting: choose the largest |a[r][k]|, r >= k
        int piv = k;
        double best = fabs(a[k][k]);
#pragma unroll
        for (int r = k + 1; r < DP; ++r) {
            const double v = fabs(a[r][k]);
            const bool better = v > best;
            best = better ? v : best;
            piv = better ? r : piv;
        }
#pragma unroll
        for (int r = k + 1; r < DP; ++r) {
            const bool sw = (piv == r);
#pragma unroll
            for (int c = 0; c < 2 * DP; ++c) {
                const double x = a[k][c], y = a[r][c];
                a[k][c] = sw ? y : x;
                a[r][c] = sw ? x : y;
            }
        }
        det = (piv != k) ? -det : det;
        const double pv = a[k][k];
        det *= pv;
        const double ip = 1.0 / pv;
#pragma unroll
        for (int r = k + 1; r < DP; ++r) {
            const double f = a[r][k] * ip;
#pragma unroll
            for (int c = k + 1; c < 2 * DP; ++c) a[r][c] = fma(-f, a[k][c], a[r][c]);
        }
    }
#pragma unroll
    for (int k = DP - 1; k >= 0; --k) {
        const double ip = 1.0 / a[k][k];
#pragma unroll
        for (int c = DP; c < 2 * DP; ++c) {
            double v = a[k][c];
#pragma unroll
            for (int r = k + 1; r < DP; ++r) v = fma(-a[k][r], a[r][c], v);
            a[k][c] = v * ip;
        }
    }
    return det;
}

// Closed-form (adjugate) solve for DP <= 3: the matrices on this path are Sigma + diag(l^2) and
// Sigma diag(.) + I with Sigma small, i.e. far from singular, so cofactor expansion is accurate to a
// few ulps and an order of magnitude shorter than elimination (the solve sits on the serial
// per-step critical path of one wavefront).  [A | RHS] -> RHS := A^-1 RHS, returns det(A).
template <int DP>
__device__ inline double adjugate_solve(double (&a)[DP][2 * DP]) {
    static_assert(DP == 2 || DP == 3, "closed form only for 2x2 / 3x3");
    double inv[DP][DP];
    double det;
    if constexpr (DP == 2) {
        det = fma(a[0][0], a[1][1], -(a[0][1] * a[1][0]));
        const double id = 1.0 / det;
        inv[0][0] = a[1][1] * id;  inv[0][1] = -a[0][1] * id;
        inv[1][0] = -a[1][0] * id; inv[1][1] = a[0][0] * id;
    } else {
        const double c00 = fma(a[1][1], a[2][2], -(a[1][2] * a[2][1]));
        const double c01 = fma(a[1][2], a[2][0], -(a[1][0] * a[2][2]));
        const double c02 = fma(a[1][0], a[2][1], -(a[1][1] * a[2][0]));
        det = fma(a[0][0], c00, fma(a[0][1], c01, a[0][2] * c02));
        const double id = 1.0 / det;
        inv[0][0] = c00 * id;
        inv[1][0] = c01 * id;
        inv[2][0] = c02 * id;
        inv[0][1] = fma(a[0][2], a[2][1], -(a[0][1] * a[2][2])) * id;
        inv[1][1] = fma(a[0][0], a[2][2], -(a[0][2] * a[2][0])) * id;
        inv[2][1] = fma(a[0][1], a[2][0], -(a[0][0] * a[2][1])) * id;
        inv[0][2] = fma(a[0][1], a[1][2], -(a[0][2] * a[1][1])) * id;
        inv[1][2] = fma(a[0][2], a[1][0], -(a[0][0] * a[1][2])) * id;
        inv[2][2] = fma(a[0][0], a[1][1], -(a[0][1] * a[1][0])) * id;
    }
    double out[DP][DP];
#pragma unroll
    for (int i = 0; i < DP; ++i)
#pragma unroll
        for (int j = 0; j < DP; ++j) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < DP; ++k) v = fma(inv[i][k], a[k][DP + j], v);
            out[i][j] = v;
        }
#pragma unroll
    for (int i = 0; i < DP; ++i)
#pragma unroll
        for (int j = 0; j < DP; ++j) a[i][DP + j] = out[i][j];
    return det;
}

template <int DP>
__device__ inline double small_solve(double (&a)[DP][2 * DP]) {
    if constexpr (DP <= 3) return adjugate_solve<DP>(a);
    else return gauss_solve_reg<DP>(a);
}

__device__ inline double norm_cdf_ref(double x, double mu, double sigma) {
    // normal_cdf of the reference (control_objects/utils/pytorch_utils.py:16-17)
    return 0.5 * (1.0 + erf((x - mu) / (sigma * 1.4142135623730951)));
}

// 1/k!
__device__ constexpr double kInvFact[kMaxTaylor + 1] = {
    1.0, 1.0, 0.5, 1.0 / 6, 1.0 / 24, 1.0 / 120, 1.0 / 720, 1.0 / 5040, 1.0 / 40320, 1.0 / 362880, 1.0 / 3628800,
    1.0 / 39916800, 1.0 / 479001600, 1.0 / 6227020800.0, 1.0 / 87178291200.0};

// kTaylorMaxArg[K] = largest c with c^(K+1)/(K+1)! * exp(2c) <= 2^-54  (computed offline, rounded down)
__device__ constexpr double kTaylorMaxArg[kMaxTaylor + 1] = {
    0.0, 1.052584e-08, 6.924801e-06, 1.908411e-04, 1.458895e-03, 5.830057e-03, 1.600533e-02, 3.463687e-02, 6.382214e-02, 1.049198e-01, 1.585847e-01, 2.249037e-01, 3.035549e-01, 3.939504e-01, 4.953495e-01};

template <int K>
__device__ inline double taylor_exp(double c) {
    double p = kInvFact[K];
#pragma unroll
    for (int k = K - 1; k >= 0; --k) p = fma(p, c, kInvFact[k]);
    return p;
}

constexpr int kTPad = 72;      // zero rows appended to every T_a: a wave may run CH <= 64 rows past the data and prefetches 4 more

// One (pair, row-chunk, column) item of the pairwise work, Taylor form.  `nrows` is wave-uniform and a
// multiple of 4; rows beyond the data are zero padding and T is zero below its diagonal, so the loop
// body carries no predication (scalar loop, loads issue ahead of the math).
//   diagonal pair : sum_i T[i][j] * ea_i * P_K(g_i . w_j)            (row record [0] = ea_i)
//   off-diagonal  : sum_i ra_i * P_K(g_i . w_j)                      (row record [1] = beta_ai ea_i)
template <int DP, int K, class REC = RowRecPacked<DP>>
__device__ inline double item_taylor(const double* rec, int nrows, const double (&w)[DP], bool diag, const double* Tp, int N) {
    constexpr int RS = REC::RS;
    constexpr int U = 4;
    double acc = 0.0;
    if (diag) {
        double tv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) tv[u] = Tp[(size_t)u * N];
        for (int it = 0; it < nrows; it += U) {
            // next group's T values: in flight during this group's math (reading past the last group only
            // touches the zero padding rows of T)
            double tn[U];
#pragma unroll
            for (int u = 0; u < U; ++u) tn[u] = Tp[(size_t)(U + u) * N];
            double cc[U], ev[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                double r[RS];
                REC::template load<(REC::EA > REC::G ? REC::EA + 1 : REC::G + DP)>(rec + u * RS, r);
                double c = r[REC::G] * w[0];
#pragma unroll
                for (int d = 1; d < DP; ++d) c = fma(r[REC::G + d], w[d], c);
                cc[u] = c;
                ev[u] = r[REC::EA];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc = fma(taylor_exp<K>(cc[u]) * ev[u], tv[u], acc);
#pragma unroll
            for (int u = 0; u < U; ++u) tv[u] = tn[u];
            rec += U * RS;
            Tp += (size_t)U * N;
        }
    } else {
        for (int it = 0; it < nrows; it += U) {
            double cc[U], rv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                double r[RS];
                REC::template load<RS>(rec + u * RS, r);
                double c = r[REC::G] * w[0];
#pragma unroll
                for (int d = 1; d < DP; ++d) c = fma(r[REC::G + d], w[d], c);
                cc[u] = c;
                rv[u] = r[REC::RA];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc = fma(taylor_exp<K>(cc[u]), rv[u], acc);
            rec += U * RS;
        }
    }
    return acc;
}

// exp(x) for the direct (fallback) evaluation: x = (64 m + j) ln2/64 + r, |r| <= ln2/128,
// exp(x) = 2^m * 2^(j/64) * (1 + r + r^2/2 + ... + r^5/120); 2^(j/64) from a 64-entry table, truncation
// 3.5e-17, total error ~1 ulp.  No overflow/underflow special-casing: arguments on this path are sums of
// log-kernel terms (<= a few units), and ldexp flushes tiny results to zero like exp does.
__device__ const double kExp2Tab[64] = {
    0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
    0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0};

__device__ inline double fast_exp(double x, const double* tab /* kExp2Tab copied to LDS */) {
    const double n = __builtin_rint(x * 0x1.71547652b82fep+6);
    double r = fma(n, -0x1.62e42fefa0000p-7, x);
    r = fma(n, -0x1.cf79abc9e3b3ap-46, r);
    const int ni = (int)n;
    const double t = tab[ni & 63];
    double q = fma(r, 0x1.1111111111111p-7, 0x1.5555555555555p-5);     // 1/120, 1/24
    q = fma(q, r, 0x1.5555555555555p-3);                                 // 1/6
    q = fma(q, r, 0.5);
    const double p = fma(q * r, r, r);                                    // e^r - 1
    return ldexp(fma(t, p, t), ni >> 6);
}

// Same item, direct form exp(ka'_i + kb'_j + g_i . w_j)  (row record [0] = ka'_i, [1] = beta_ai).
template <int DP, class REC = RowRecPacked<DP>>
__device__ inline double item_exp(const double* rec, int nrows, const double (&w)[DP], double kbj, bool diag,
                                  const double* Tp, int N, const double* tab) {
    constexpr int RS = REC::RS;
    constexpr int U = 4;
    double acc = 0.0;
    if (diag) {
        for (int it = 0; it < nrows; it += U) {
            double tv[U], aa[U];
#pragma unroll
            for (int u = 0; u < U; ++u) tv[u] = Tp[(size_t)u * N];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                double r[RS];
                REC::template load<(REC::EA > REC::G ? REC::EA + 1 : REC::G + DP)>(rec + u * RS, r);
                double arg = r[REC::EA] + kbj;
#pragma unroll
                for (int d = 0; d < DP; ++d) arg = fma(r[REC::G + d], w[d], arg);
                aa[u] = arg;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc = fma(fast_exp(aa[u], tab), tv[u], acc);
            rec += U * RS;
            Tp += (size_t)U * N;
        }
    } else {
        for (int it = 0; it < nrows; it += U) {
            double aa[U], bv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                double r[RS];
                REC::template load<RS>(rec + u * RS, r);
                double arg = r[REC::EA] + kbj;
#pragma unroll
                for (int d = 0; d < DP; ++d) arg = fma(r[REC::G + d], w[d], arg);
                aa[u] = arg;
                bv[u] = r[REC::RA];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc = fma(fast_exp(aa[u], tab), bv[u], acc);
            rec += U * RS;
        }
    }
    return acc;
}

// Two columns per lane (j, j + 1): the row operands (ev_i, g_i) are LDS broadcasts whose return bandwidth, not the
// VALU, bounds the one-column loop at small D (32 bytes per lane and row for 12 FMAs); serving two columns per read
// halves that traffic.  Two rows x two columns in flight = the same four independent chains as the one-column loop.
// U = rows per trip = depth of the T prefetch.  With the tables in L2 (configs 1-3) two rows ahead are enough and the
// shorter trip is 2 % faster; at config 4 (N = 1000, D = 4) the loads come from the Infinity Cache and four rows ahead gain
// 5 % (115.2 -> 109.2 ms); eight rows spill and lose.
template <int DP, int K, int U, class REC = RowRecAligned<DP>>
__device__ inline void item_taylor2(const double* rec, int nrows, const double (&w0)[DP], const double (&w1)[DP], bool diag,
                                    const double* Tp, int N, double& acc0, double& acc1) {
    constexpr int RS = REC::RS;
    acc0 = 0.0;
    acc1 = 0.0;
    if (diag) {
        double t0[U], t1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { t0[u] = Tp[(size_t)u * N]; t1[u] = Tp[(size_t)u * N + 1]; }
        for (int it = 0; it < nrows; it += U) {
            double n0[U], n1[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { n0[u] = Tp[(size_t)(U + u) * N]; n1[u] = Tp[(size_t)(U + u) * N + 1]; }
            double c0[U], c1[U], ev[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                double r[RS];
                REC::template load<(REC::EA > REC::G ? REC::EA + 1 : REC::G + DP)>(rec + u * RS, r);
                double a = r[REC::G] * w0[0], b = r[REC::G] * w1[0];
#pragma unroll
                for (int d = 1; d < DP; ++d) { a = fma(r[REC::G + d], w0[d], a); b = fma(r[REC::G + d], w1[d], b); }
                c0[u] = a; c1[u] = b;
                ev[u] = r[REC::EA];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                acc0 = fma(taylor_exp<K>(c0[u]) * ev[u], t0[u], acc0);
                acc1 = fma(taylor_exp<K>(c1[u]) * ev[u], t1[u], acc1);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { t0[u] = n0[u]; t1[u] = n1[u]; }
            rec += U * RS;
            Tp += (size_t)U * N;
        }
    } else {
        for (int it = 0; it < nrows; it += U) {
            double c0[U], c1[U], rv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                double r[RS];
                REC::template load<RS>(rec + u * RS, r);
                double a = r[REC::G] * w0[0], b = r[REC::G] * w1[0];
#pragma unroll
                for (int d = 1; d < DP; ++d) { a = fma(r[REC::G + d], w0[d], a); b = fma(r[REC::G + d], w1[d], b); }
                c0[u] = a; c1[u] = b;
                rv[u] = r[REC::RA];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                acc0 = fma(taylor_exp<K>(c0[u]), rv[u], acc0);
                acc1 = fma(taylor_exp<K>(c1[u]), rv[u], acc1);
            }
            rec += U * RS;
        }
    }
}

template <int DP, class REC = RowRecAligned<DP>>
__device__ inline void item_exp2(const double* rec, int nrows, const double (&w0)[DP], const double (&w1)[DP], double kb0,
                                 double kb1, bool diag, const double* Tp, int N, const double* tab, double& acc0, double& acc1) {
    constexpr int RS = REC::RS;
    constexpr int U = 2;
    acc0 = 0.0;
    acc1 = 0.0;
    for (int it = 0; it < nrows; it += U) {
        double a0[U], a1[U], wt0[U], wt1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            double r[RS];
            REC::template load<RS>(rec + u * RS, r);
            double x = r[REC::EA] + kb0, y = r[REC::EA] + kb1;
#pragma unroll
            for (int d = 0; d < DP; ++d) { x = fma(r[REC::G + d], w0[d], x); y = fma(r[REC::G + d], w1[d], y); }
            a0[u] = x; a1[u] = y;
            wt0[u] = diag ? Tp[(size_t)u * N] : r[REC::RA];
            wt1[u] = diag ? Tp[(size_t)u * N + 1] : r[REC::RA];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc0 = fma(fast_exp(a0[u], tab), wt0[u], acc0);
            acc1 = fma(fast_exp(a1[u], tab), wt1[u], acc1);
        }
        rec += U * RS;
        Tp += (size_t)U * N;
    }
}

// Sums of 8 per-lane values over the wavefront in ~1/3 of the instructions of 8 separate reductions: at every step lanes
// that differ in one index bit exchange the half of their values the partner keeps, so the value count halves while the
// span doubles (8 -> 4 -> 2 -> 1 over lane bits 0, 1, 2), then the single survivor is summed over lane bits 3, 4, 5.
// Lane l ends with the total of value  m(l) = 4 (l & 1) + 2 ((l >> 1) & 1) + ((l >> 2) & 1).  Fixed order.
__device__ inline double wave_sum8(const double (&v)[8], int lane) {
    auto xchg = [](double x, int mask) { return __shfl_xor(x, mask, 64); };
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    double r4[4], r2[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double keep = b0 ? v[k + 4] : v[k], send = b0 ? v[k] : v[k + 4];
        r4[k] = keep + xchg(send, 1);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const double keep = b1 ? r4[k + 2] : r4[k], send = b1 ? r4[k] : r4[k + 2];
        r2[k] = keep + xchg(send, 2);
    }
    double r = (b2 ? r2[1] : r2[0]) + xchg(b2 ? r2[0] : r2[1], 4);
    r += xchg(r, 8);
    r += xchg(r, 16);
    r += xchg(r, 32);
    return r;
}

// Sums over the wavefront of 16 (8) values per lane without the LDS crossbar: the halving "transpose" steps ride on
// v_permlane32_swap / v_permlane16_swap (gfx950: lanes 0-31 <-> 32-63, even <-> odd rows of 16) and on DPP row rotations with
// bank masks (lane bits 3 and 2) -- one exchange hands over the half a lane gives up AND brings in the partner's half of what it
// keeps, no selects -- then two quad_perm butterflies.  Lane l ends with the total of value (l >> 2) [& 7].  57 VALU instructions for
// 16 values against ~70 plus six dependent ds_bpermute round trips for the 8 values of wave_sum8.  Fixed order.
__device__ inline double dbl_of(unsigned lo, unsigned hi) { return __hiloint2double((int)hi, (int)lo); }

template <int KIND>                     // 32: lane bit 5, 16: lane bit 4
__device__ inline double swap_add(double a, double b) {
    const unsigned a0 = (unsigned)__double2loint(a), a1 = (unsigned)__double2hiint(a);
    const unsigned b0 = (unsigned)__double2loint(b), b1 = (unsigned)__double2hiint(b);
    if constexpr (KIND == 32) {
        const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        return dbl_of(r0[0], r1[0]) + dbl_of(r0[1], r1[1]);
    } else {
        const auto r0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
        const auto r1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
        return dbl_of(r0[0], r1[0]) + dbl_of(r0[1], r1[1]);
    }
}

template <int CTRL, int BANKS>
__device__ inline double dpp_merge(double old, double src) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, 0xf, BANKS, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, 0xf, BANKS, false);
    return __hiloint2double(hi, lo);
}

// lanes with bit 3 (2) clear keep `a`, the others `b`; both get the partner's (lane ^ 8, lane ^ 4) share of what they keep
__device__ inline double rot_add8(double a, double b) { return dpp_merge<0x128, 0x3>(b, a) + dpp_merge<0x128, 0xc>(a, b); }
__device__ inline double rot_add4(double a, double b) { return dpp_merge<0x12c, 0x5>(b, a) + dpp_merge<0x124, 0xa>(a, b); }

__device__ inline double quad_total(double r) {
    r += dpp_merge<0x4e, 0xf>(r, r);          // quad_perm [2, 3, 0, 1]
    r += dpp_merge<0xb1, 0xf>(r, r);          // quad_perm [1, 0, 3, 2]
    return r;
}

__device__ inline double wave_reduce16(const double (&v)[16]) {
    double w8[8], w4[4], w2[2];
#pragma unroll
    for (int k = 0; k < 8; ++k) w8[k] = swap_add<32>(v[k], v[k + 8]);
#pragma unroll
    for (int k = 0; k < 4; ++k) w4[k] = swap_add<16>(w8[k], w8[k + 4]);
#pragma unroll
    for (int k = 0; k < 2; ++k) w2[k] = rot_add8(w4[k], w4[k + 2]);
    return quad_total(rot_add4(w2[0], w2[1]));
}

__device__ inline double wave_reduce8(const double (&v)[8]) {
    double w4[4], w2[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) w4[k] = swap_add<16>(v[k], v[k + 4]);
#pragma unroll
    for (int k = 0; k < 2; ++k) w2[k] = rot_add8(w4[k], w4[k + 2]);
    const double r = quad_total(rot_add4(w2[0], w2[1]));
    return swap_add<32>(r, r);
}

// ------------------------------------------------------------------------------------------
// Separable moments for three state dimensions with the monomial structure known at compile time.
// Bands of at most 16 monomials (ranges of the x0 exponent i and, where needed, of the x1 exponent j); fewer, larger
// bands for the low degrees so that their items are not all set-up and reduction:
//   KS <= 3: {i = 0}, {i >= 1}                      KS = 4: {i = 0}, {i = 1}, {i >= 2}
//   KS = 5: {i = 0, j <= 1}, {i = 0, j >= 2}, {i = 1}, {i = 2}, {i >= 3}
//   KS = 6: {i = 0, j <= 1}, {i = 0, j >= 2}, {i = 1, j <= 1}, {i = 1, j >= 2}, {i = 2}, {i = 3}, {i >= 4}
// Inside a band the local order is i, j, k ascending; the canonical (graded, i then j descending) index the host tables
// use is  cum(deg - 1) + T(deg - i) + (deg - i - j),  deg = i + j + k,  cum(n) = (n+1)(n+2)(n+3)/6,  T(n) = n(n+1)/2.
constexpr int sep3_bands(int KS) { return KS <= 3 ? 2 : (KS == 4 ? 3 : (KS == 5 ? 5 : 7)); }
constexpr int sep3_i0(int B, int KS) {
    if (KS <= 4) return B;
    if (KS == 5) return B < 2 ? 0 : B - 1;
    return B < 2 ? 0 : (B < 4 ? 1 : B - 2);
}
constexpr int sep3_i1(int B, int KS) { return B == sep3_bands(KS) - 1 ? KS : sep3_i0(B, KS); }
constexpr int sep3_j0(int B, int KS) {
    if (KS == 5) return B == 1 ? 2 : 0;
    if (KS >= 6) return (B == 1 || B == 3) ? 2 : 0;
    return 0;
}
constexpr int sep3_j1(int B, int KS) {
    if (KS == 5) return B == 0 ? 1 : KS;
    if (KS >= 6) return (B == 0 || B == 2) ? 1 : KS;
    return KS;
}
constexpr int sep3_count(int B, int KS) {
    int n = 0;
    if (B >= sep3_bands(KS)) return 0;
    for (int i = sep3_i0(B, KS); i <= sep3_i1(B, KS) && i <= KS; ++i)
        for (int j = sep3_j0(B, KS); j <= sep3_j1(B, KS) && j <= KS - i; ++j) n += KS - i - j + 1;
    return n;
}
template <int KS, int B>
struct Sep3Canon {
    int v[sep3_count(B, KS) > 0 ? sep3_count(B, KS) : 1];
    constexpr Sep3Canon() : v{} {
        int n = 0;
        for (int i = sep3_i0(B, KS); i <= sep3_i1(B, KS) && i <= KS && B < sep3_bands(KS); ++i)
            for (int j = sep3_j0(B, KS); j <= sep3_j1(B, KS) && j <= KS - i; ++j)
                for (int kk = 0; kk <= KS - i - j; ++kk) {
                    const int deg = i + j + kk;
                    const int cum = deg * (deg + 1) * (deg + 2) / 6;                 // monomials of degree < deg
                    v[n++] = cum + (deg - i) * (deg - i + 1) / 2 + (deg - i - j);
                }
    }
};

// One (side, band) item: lanes own points; every monomial of the band costs one FMA per point (x1^j, x2^k tabulated per
// point, wt x0^i x1^j formed once per (i, j)); the lane partials are reduced 8 at a time and scattered to the
// canonical positions of this (pair, side) in `mom`.
// (side 0: rows, x = g_i from the row records, weight rec[1]; side 1: columns, x = nu_j / l_b^2, weight kb)
template <int KS, int B>
__device__ inline void sep3_band(int lane, int N, int side, const double* rec0, int RS, const double* kb, const double* nu,
                                 const double* il, double* mom) {
    constexpr int NBm = sep3_count(B, KS);
    if constexpr (NBm > 0) {
        constexpr int NP = (NBm + 7) & ~7;
        constexpr int I0 = sep3_i0(B, KS), I1 = sep3_i1(B, KS) < KS ? sep3_i1(B, KS) : KS;
        constexpr int J0 = sep3_j0(B, KS), J1 = sep3_j1(B, KS);
        double acc[NP];
#pragma unroll
        for (int m = 0; m < NP; ++m) acc[m] = 0.0;
        for (int pt = lane; pt < N; pt += 64) {
            double x0, x1, x2, wt;
            if (side == 0) {
                // lanes own points: per-lane 16-byte reads at the record stride of 48 bytes are conflict-free (the start banks of
                // a 16-lane group are the 16 multiples of 4)
                double r[RowRecAligned<3>::RS];
                RowRecAligned<3>::load<RowRecAligned<3>::RS>(rec0 + (size_t)pt * RS, r);
                wt = r[RowRecAligned<3>::RA]; x0 = r[0]; x1 = r[1]; x2 = r[2];
            } else {
                wt = kb[pt];
                x0 = nu[pt] * il[0]; x1 = nu[N + pt] * il[1]; x2 = nu[2 * N + pt] * il[2];
            }
            double py[KS + 1], pz[KS + 1];
            py[0] = 1.0; pz[0] = 1.0;
#pragma unroll
            for (int e = 1; e <= KS; ++e) { py[e] = py[e - 1] * x1; pz[e] = pz[e - 1] * x2; }
            double wx = wt;
#pragma unroll
            for (int e = 0; e < I0; ++e) wx *= x0;
            int n = 0;
#pragma unroll
            for (int i = I0; i <= I1; ++i) {
#pragma unroll
                for (int j = J0; j <= (J1 < KS - i ? J1 : KS - i); ++j) {
                    const double aij = wx * py[j];
#pragma unroll
                    for (int kk = 0; kk <= KS - i - j; ++kk) { acc[n] = fma(aij, pz[kk], acc[n]); ++n; }
                }
                wx *= x0;
            }
        }
        // lane partials -> totals, sixteen (eight) at a time on the permlane / DPP exchange network (wave_reduce16: 57 vector
        // instructions, no LDS round trip; the ds_bpermute butterflies of wave_sum8 took ~70 instructions and six dependent LDS
        // round trips per EIGHT values -- the reductions were more than half of a low-degree item)
        static constexpr Sep3Canon<KS, B> canon{};
        const int mq = lane >> 2;
#pragma unroll
        for (int g16 = 0; g16 + 16 <= NP; g16 += 16) {
            const double(&grp)[16] = *reinterpret_cast<const double(*)[16]>(&acc[g16]);
            const double tot = wave_reduce16(grp);
            if ((lane & 3) == 0 && g16 + mq < NBm) mom[canon.v[g16 + mq]] = tot;
        }
        if constexpr (NP % 16 != 0) {
            constexpr int g8 = NP - 8;
            const double(&grp)[8] = *reinterpret_cast<const double(*)[8]>(&acc[g8]);
            const double tot = wave_reduce8(grp);
            if ((lane & 3) == 0 && lane < 32 && g8 + mq < NBm) mom[canon.v[g8 + mq]] = tot;
        }
    }
}

// Not inlined: the band code wants ~90 VGPRs of its own, next to the ~60 the work-queue loop keeps live.
template <int KS>
__device__ __attribute__((noinline)) void sep3_item(int band, int lane, int N, int side, const double* rec0, int RS,
                                                     const double* kb, const double* nu, const double* il, double* mom) {
    switch (band) {
        case 0: sep3_band<KS, 0>(lane, N, side, rec0, RS, kb, nu, il, mom); break;
        case 1: sep3_band<KS, 1>(lane, N, side, rec0, RS, kb, nu, il, mom); break;
        case 2: sep3_band<KS, 2>(lane, N, side, rec0, RS, kb, nu, il, mom); break;
        case 3: sep3_band<KS, 3>(lane, N, side, rec0, RS, kb, nu, il, mom); break;
        case 4: sep3_band<KS, 4>(lane, N, side, rec0, RS, kb, nu, il, mom); break;
        case 5: sep3_band<KS, 5>(lane, N, side, rec0, RS, kb, nu, il, mom); break;
        default: sep3_band<KS, 6>(lane, N, side, rec0, RS, kb, nu, il, mom); break;
    }
}

__device__ inline int wave_max_i32(int v) {
    auto step = [&](auto ctrl, auto rmask) {
        const int o = __builtin_amdgcn_update_dpp(0, v, decltype(ctrl)::value, decltype(rmask)::value, 0xf, true);
        v = o > v ? o : v;
    };
    step(std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});
    step(std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});
    step(std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});
    step(std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});
    step(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});
    step(std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});
    return __builtin_amdgcn_readlane(v, 63);        // values are >= 0, so the zero fill is neutral
}

// ------------------------------------------------------------------------------------------
// DX = exact state dimension known at compile time (0: runtime p.D <= DP): folds every D-dependent
// offset and small loop, which is what keeps the 128-VGPR budget of a 1024-thread workgroup.
// TILED = one step of the per-candidate part of the batch-major path (pair_tile_kernel.h): the horizon slice
// [p.t_begin, p.t_end) starts from the state stored in the trajectory arrays, the diagonal output pairs are not
// processed here -- their N x N sums arrive as per-tile partial sums in p.tile_part and are added in a fixed order.
// CL = the few-candidate cooperative form (the reference's own regime: restarts_optim 1-2, one candidate per objective
// evaluation, gp_mpc_controller.py:125-141): p.cluster workgroups share ONE candidate.  Every member keeps the whole state and runs
// the small algebra, the state update and the horizon loop itself; the step's work items (element-wise N x N items, separable
// pairs, mean sums) have a static owner among the members, a member runs the per-point pass only for the output pairs it owns
// items of, and the items' results travel once per step as tagged granules through L2 / the fabric (two 8-byte agent-scope atomic
// stores {tag | low word}, {tag | high word} per value: each half is single-copy atomic, a reader accepts a value when both tags
// are this step's -- no fences, MI355X_MICROARCH.md "handoff-1to1").  Every value is formed by the same instruction sequence as in
// the plain kernel and summed in the same order: the trajectory is bit-identical to the one-workgroup result.
template <int DP, int NT, int DX, bool C2, bool TILED = false, bool CL = false>
__global__ __launch_bounds__(NT) void rollout_kernel(const RolloutArgs p) {
    static_assert(!TILED || (DP <= 4 && NT >= 256), "the batch-major path is built for D <= 4");
    static_assert(!CL || (!TILED && DP <= 4 && NT >= 256), "the cooperative form is built for D <= 4");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if constexpr (TILED) {
        // batch-major path: this kernel only takes the candidates point_pass_kernel left (element-wise off-diagonal pairs)
        if (p.slow[blockIdx.x] != p.t_begin + 1) return;
    }
    constexpr int NW = NT / kWave;
    using REC = RowRecAligned<DP>;          // row record: g_i (DP) | ea_i or ka'_i | ra_i or beta_ai | pad, 16-byte aligned
    constexpr int RS = REC::RS;
    const int tid = threadIdx.x;
    // cooperative form: workgroup b is member (b / 8) % cluster of candidate (b % 8) + 8 (b / (8 cluster)) -- the members of a
    // candidate are 8 apart in dispatch order, i.e. on one XCD where the dispatcher places block b on XCD b % 8 (speed only)
    // (cluster_debug 8, a test hook: members of a candidate on CONSECUTIVE blocks instead, i.e. spread over the XCDs -- the
    //  placement-independent form of the exchange is then what runs, across L2s for real)
    const int CS = CL ? p.cluster : 1;
    const bool spread = CL && (p.cl_dbg & 8);
    const int member = CL ? (spread ? (int)blockIdx.x % CS : (int)(blockIdx.x >> 3) % CS) : 0;
    const int c = CL ? (spread ? (int)blockIdx.x / CS : (int)(blockIdx.x & 7) + 8 * ((int)(blockIdx.x >> 3) / CS)) : (int)blockIdx.x;
    if constexpr (CL) { if (c >= p.B) return; }
    const int D = (DX > 0) ? DX : p.D;
    const int N = p.N, A = p.A, E = p.E, H = p.H, G = p.G, CM = p.CM;
    const int P = TILED ? D * (D - 1) / 2 : D * (D + 1) / 2;      // output pairs whose N x N work is done here
    const int DA = D + A;
    const int LD = 2 * D;                   // row stride of an augmented block
    const int NC = C2 ? (N + 1) / 2 : N;    // column units per row chunk: columns, or pairs of adjacent columns
    const int wpp = (p.RC * NC + 63) / 64;  // work-item slots per output pair
    const int SD2 = rnd2(D * D);

    const Layout L = make_layout(N, D, A, E, G, DP, wpp, CM, p.CH, H * A, p.x_in_lds != 0, CL, CL ? p.cl_slots : 0);     // the host sized it with the same wpp
    const int NR = N + p.CH;                // rows per pair in the row-record array (data + zero padding)
    double* s_mu = smem + L.mu;
    double* s_Sig2 = smem + L.Sig;
    double* s_m = smem + L.m;
    double* s_M = smem + L.M;
    double* s_cc = smem + L.cc;
    double* s_s1 = smem + L.s1;             // [a][0] = sum lb, [a][1+d] = sum lb * nu_d
    double* s_Vs = smem + L.Vs;             // [k][a]
    double* s_Sp = smem + L.Sp;
    double* s_rdet = smem + L.rdet;
    [[maybe_unused]] double* s_rdiag = smem + L.misc;      // TILED: 1 / sqrt(det R_aa) [0, D) and the tile sums [4, 4 + D)
    double* s_aug = smem + L.aug;           // problems [0, D): mean part, [D, D+G): pairs of the group
    double* s_part = smem + L.part;
    double* s_mom = smem + L.mom;           // [gq][side][CM]
    int* s_pa = reinterpret_cast<int*>(smem + L.ints);
    int* s_pb = s_pa + P;
    int* s_K = s_pb + P;                    // per pair of the group: Taylor degree, 0 = direct exp
    int* s_counter = s_K + G;
    int* s_noff = s_counter + 1;            // off-diagonal pairs of the current group: count and their slots
    int* s_off = s_noff + 1;
    int* s_mcum = s_off + G;                // monomials of degree <= k (copy of the launch argument: LDS instead of a kernarg load on the serial path)
    int* s_tri = s_mcum + 16;               // diagonal pairs: column units of row chunks < r that can hold an element i <= j
    int* s_nslot = s_tri + tri_ints(N);     // work-item slots of each pair of the group that hold work at this step
    int* s_nitems = s_nslot + G;            // length of the step's work-item list
    // The work-item list of the step (built by one thread beside the per-point pass, read by the queue of P3): only slots
    // that hold work -- a diagonal pair uses the slots of its triangle, a separable pair one slot per (side, monomial band) --
    // long items (element-wise pairs) first, the short mean sums last.  Entry = pair slot index gq * wpp + slot, or
    // 0xffff - a for the mean sums of output a.  (Until round 5 the queue walked all G * wpp slots: at config 2, 36 of 69
    // pulls per step found nothing to do, each after the lane set-up of an item.)
    unsigned short* s_items = reinterpret_cast<unsigned short*>(
        smem + L.ints + rnd2((2 * P + 3 * G + 8 + 16 + tri_ints(N) + 1) / 2));

    // Lane map of a diagonal pair's work items (slot, lane) -> (row chunk | column unit << 8), -1 = no element, and the row count
    // of each slot: the triangle's enumeration does not depend on the state, so the per-item binary search over s_tri and the
    // wavefront maximum of the lanes' row counts are done ONCE per launch (at config 2 they were ~70 of an item's ~250 set-up
    // instructions, 18 items per step).
    // cooperative form
    [[maybe_unused]] double* s_sepv = smem + L.cl;                                          // totals of the separable pairs
    [[maybe_unused]] int* s_ord = reinterpret_cast<int*>(smem + L.cl + rnd2(G));            // ordinal of a pair among the diagonal / off-diagonal pairs
    [[maybe_unused]] int* s_fail = reinterpret_cast<int*>(smem + L.cl + rnd2(G) + rnd2((G + 1) / 2));
    [[maybe_unused]] unsigned char* s_needw = reinterpret_cast<unsigned char*>(smem + L.cl + rnd2(G) + rnd2((G + 1) / 2) + 2);
    [[maybe_unused]] unsigned char* s_own = s_needw + kClusterScratch * 8;      // owner of item slot gq * wpp + slot (element-wise form)
    [[maybe_unused]] unsigned char* s_sepown = s_own + G * wpp;                 // owner of pair gq when it is separable
    [[maybe_unused]] unsigned char* s_elneed = s_sepown + G;                    // 1: this member owns a slot of pair gq's element-wise form
    [[maybe_unused]] unsigned char* s_meanown = s_elneed + G;                   // owner of the mean sums of output a
    [[maybe_unused]] unsigned short* s_mine = reinterpret_cast<unsigned short*>(       // this member's items of the step, [0] = their number
        smem + L.cl + rnd2(G) + rnd2((G + 1) / 2) + 2 + kClusterScratch + rnd2((G * wpp + 2 * G + 16 + 7) / 8));
    [[maybe_unused]] int* s_slot = reinterpret_cast<int*>(smem + L.cl + rnd2(G) + rnd2((G + 1) / 2) + 2 + kClusterScratch +
                                                          rnd2((G * wpp + 2 * G + 16 + 7) / 8) + rnd2((G * wpp + 16 + 4 + 3) / 4));
    [[maybe_unused]] unsigned long long* s_mymask = reinterpret_cast<unsigned long long*>(     // element-wise slots of pair gq this member owns
        smem + L.cl + rnd2(G) + rnd2((G + 1) / 2) + 2 + kClusterScratch + rnd2((G * wpp + 2 * G + 16 + 7) / 8) +
        rnd2((G * wpp + 16 + 4 + 3) / 4) + rnd2((G + 1) / 2));
    const int GL = CL ? p.cl_slots : G;             // LDS slots of per-pair records

    [[maybe_unused]] unsigned long long* xbuf_uc = CL ? p.xch_uc + (size_t)c * 4 * p.xch_n : nullptr;    // the same in UNCACHED memory: prologue, and the steps of members on several XCDs
    [[maybe_unused]] unsigned long long* xbuf = CL ? p.xch + (size_t)c * 4 * p.xch_n : nullptr;    // [buffer][value][2]
    [[maybe_unused]] const int x_sep = G * wpp, x_mean = G * wpp + G;                       // value index: item slots | separable pairs | mean sums
    [[maybe_unused]] const int x_xcc = x_mean + D * (D + 1);                                // ... | the members' XCD ids (prologue)
    [[maybe_unused]] int* s_lmap = reinterpret_cast<int*>(smem + L.lmap);
    constexpr int LMS = CL ? kLaneMapSlotsCluster : kLaneMapSlots;
    [[maybe_unused]] int* s_lrows = s_lmap + (wpp < LMS ? wpp : LMS) * 64;
    double* ppbase = smem;                  // per-point arrays live in LDS (large N: rollout_stream_kernel.h)
    double* a_nu = ppbase + L.nu;           // [d][p]
    double* a_lb = ppbase + L.lb;           // [a][p]
    double* a_rows = ppbase + L.rows;       // [gq][p][RS]
    double* a_kb = ppbase + L.kb;           // [gq][p]

    double* c_ils2 = smem + L.c_ils2;
    double* c_logvar = smem + L.c_logvar;
    double* c_var = smem + L.c_var;
    double* c_xr = smem + L.c_xr;
    double* c_act = smem + L.c_act;
    double* c_exptab = smem + L.c_exptab;
    const double* Xs = p.x_in_lds ? (smem + L.c_X) : p.Xt;     // X^T (E, N): LDS copy or HBM/L2
    double* c_monow = smem + L.c_monow;
    int* c_monoe = reinterpret_cast<int*>(smem + L.c_monoe);
    const double* act = p.actions + (size_t)c * H * A;

    [[maybe_unused]] int t_dbg = -1;
#if defined(GPMPC_PROF_ON)
    long long prof_x[3] = {0, 0, 0};
    long long prof_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long prof_last = __builtin_readcyclecounter();
    long long prof_wall[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long prof_wlast = wall_clock64();
#endif
    // ---- init -----------------------------------------------------------------------
    if constexpr (TILED) {
        for (int i = tid; i < D; i += NT) { s_mu[i] = p.mu_out[((size_t)c * (H + 1) + p.t_begin) * D + i]; c_logvar[i] = p.logvar[i]; c_var[i] = p.var[i]; }
        for (int i = tid; i < D * D; i += NT) s_Sig2[i] = p.Sig_out[((size_t)c * (H + 1) + p.t_begin) * D * D + i];
    }
    {
        // Read-only tables.  The first element (X^T: the first two) of every table per thread with ALL loads in flight together, no
        // branch between them (lanes past a table's end read its first element and drop it): the memory counter is in-order, and one
        // `for (i = tid; ...) lds[i] = table[i]` loop after the other is one global round trip after the other -- ten of them
        // were most of this phase at few candidates (round 6).
        const bool inl = p.act_inline_n > 0;
        const int CMc = CM > 0 ? CM : 0;
        const double* mw = CMc > 0 ? p.mono_w : p.ils2;
        const int* me = CMc > 0 ? p.mono_exp : reinterpret_cast<const int*>(p.ils2);
        const int im = tid < CMc ? tid : 0;
        const int nX = p.x_in_lds ? E * N : 0;
        __builtin_amdgcn_sched_barrier(0);
        const double v_lv = p.logvar[tid < D ? tid : 0], v_var = p.var[tid < D ? tid : 0];
        const double v_ils = p.ils2[tid < D * E ? tid : 0], v_xr = p.xrange[tid < 2 * E ? tid : 0];
        const double v_act = act[tid < H * A ? tid : 0];
        const double v_exp = kExp2Tab[tid & 63];
        const double v_mw = mw[im];
        const int e0 = me[im * 4], e1 = me[im * 4 + 1], e2 = me[im * 4 + 2], e3 = me[im * 4 + 3];
        const double v_x0 = p.Xt[tid < nX ? tid : 0], v_x1 = p.Xt[tid + NT < nX ? tid + NT : 0];
        [[maybe_unused]] const double v_mu0 = p.mu0[tid < D ? tid : 0], v_S0 = p.S0[tid < D * D ? tid : 0];      // (D * D <= 256 <= NT)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!TILED) {
            if (tid < D) s_mu[tid] = v_mu0;
            if (tid < D * D) s_Sig2[tid] = v_S0;
        }
        if (tid < D) { c_logvar[tid] = v_lv; c_var[tid] = v_var; }
        if (tid < D * E) c_ils2[tid] = v_ils;
        if (tid < 2 * E) c_xr[tid] = v_xr;
        if (tid < 64) c_exptab[tid] = v_exp;
        if (tid < CMc) { c_monow[tid] = v_mw; c_monoe[tid] = e0 | (e1 << 8) | (e2 << 16) | (e3 << 24); }
        if (tid < nX) smem[L.c_X + tid] = v_x0;
        if (tid + NT < nX) smem[L.c_X + tid + NT] = v_x1;
        if (inl) {            // one sequence from the host: in the argument block; a copy for the gradient's kernels
            for (int i = tid; i < H * A; i += NT) {
                const double v = p.act_inline[i < kInlineActs ? i : 0];
                c_act[i] = v;
                if (!CL || member == 0) p.act_store[i] = v;
            }
        } else {
            if (tid < H * A) c_act[tid] = v_act;
            for (int i = tid + NT; i < H * A; i += NT) c_act[i] = act[i];
        }
        // (what is left of tables longer than the workgroup)
        for (int i = tid + NT; i < D; i += NT) { c_logvar[i] = p.logvar[i]; c_var[i] = p.var[i]; }
        for (int i = tid + NT; i < D * E; i += NT) c_ils2[i] = p.ils2[i];
        for (int i = tid + NT; i < 2 * E; i += NT) c_xr[i] = p.xrange[i];
        for (int i = tid + 2 * NT; i < nX; i += NT) smem[L.c_X + i] = p.Xt[i];
        for (int i = tid + NT; i < CMc; i += NT) {
            c_monow[i] = p.mono_w[i];
            c_monoe[i] = p.mono_exp[i * 4] | (p.mono_exp[i * 4 + 1] << 8) | (p.mono_exp[i * 4 + 2] << 16) | (p.mono_exp[i * 4 + 3] << 24);
        }
    }
    for (int i = tid; i < GL * p.CH * RS; i += NT) {
        const int gq = i / (p.CH * RS), k = i - gq * (p.CH * RS);
        a_rows[((size_t)gq * NR + N) * RS + k] = 0.0;                      // zero padding rows
    }
    if (tid >= 64 && tid < 80) s_mcum[tid - 64] = p.mono_cum[tid - 64];
    if (tid == 0) {
        int q = 0;
        [[maybe_unused]] int nd_ = 0, no_ = 0;
        for (int a = 0; a < D; ++a)
            for (int b = TILED ? a + 1 : a; b < D; ++b) {
                s_pa[q] = a; s_pb[q] = b;
                if constexpr (CL) s_ord[q] = (a == b) ? nd_++ : no_++;
                ++q;
            }
        if constexpr (CL) *s_fail = 0;
        // a column unit (one column, or the pair (2 jc, 2 jc + 1)) is useful for row chunk r if its last column >= r CH
        int run = 0;
        for (int r = 0; r <= p.RC; ++r) {
            s_tri[r] = run;
            const int first = C2 ? (r * p.CH) / 2 : r * p.CH;          // first useful unit of chunk r
            run += (first < NC) ? NC - first : 0;
        }
    }
    __syncthreads();
#if !defined(GPMPC_NO_LANEMAP)
    bool use_lmap = false;
    if constexpr (!TILED) {
        const int wtri = (s_tri[p.RC] + 63) >> 6;
        use_lmap = wtri <= LMS && wtri <= wpp;
        if (use_lmap) {
            for (int slot = tid >> 6; slot < wtri; slot += NW) {
                const int flat = slot * 64 + (tid & 63);
                const bool valid = flat < s_tri[p.RC];
                int lo = 0, hi = p.RC;                         // invariant: s_tri[lo] <= flat < s_tri[hi]
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (s_tri[mid] <= flat) lo = mid; else hi = mid;
                }
                const int r = valid ? lo : 0;
                const int first = C2 ? (r * p.CH) / 2 : r * p.CH;
                const int jc = valid ? first + (flat - s_tri[r]) : 0;
                const int j = C2 ? 2 * jc : jc;
                const int jl = (C2 && (j + 1 < N)) ? j + 1 : j;
                const int i0 = r * p.CH;
                int i1 = i0 + p.CH;
                if (i1 > N) i1 = N;
                if (i1 > jl + 1) i1 = jl + 1;
                const int len = (valid && i1 > i0) ? (i1 - i0) : 0;
                const int nrows = (wave_max_i32(len) + 3) & ~3;
                s_lmap[flat] = valid ? (r | (jc << 8)) : -1;
                if ((tid & 63) == 0) s_lrows[slot] = nrows;
            }
        }
        __syncthreads();
    }
#else
    constexpr bool use_lmap = false;
#endif
    GPMPC_TRACE(1);
    if (!TILED && member == 0) {
        for (int i = tid; i < D; i += NT) p.mu_out[((size_t)c * (H + 1)) * D + i] = s_mu[i];
        for (int i = tid; i < D * D; i += NT) p.Sig_out[((size_t)c * (H + 1)) * D * D + i] = s_Sig2[i];
    }

    // ---- cooperative form: static owners of the step's work items ------------------------------------------------------
    // diagonal pairs (always element-wise): their D * wtri items in order, an equal contiguous share per member; an off-diagonal
    // pair in element-wise form: its wpp slots spread over all members (rotated by the pair's ordinal); a separable pair as a
    // whole and a mean-sum item: one member each, counted down from the last member.
    [[maybe_unused]] const int wtri_c = CL ? (s_tri[p.RC] + 63) >> 6 : 0;
    if constexpr (CL) {
        // the owners do not depend on the state: tabulated once per launch (the divisions by run-time values cost ~40 vector
        // instructions each -- per step and wavefront they were ~5 k cycles)
        const ClusterMap& cmap = p.cmap;                 // planned on the host (run-time divisions and searches cost tens of us here)
        for (int i = tid; i < P * wpp; i += NT) {
            const int gq = i / wpp, slot = i - gq * wpp, ord = s_ord[gq];
            int o;
            if (s_pa[gq] == s_pb[gq]) o = slot < wtri_c ? cmap.diag_owner(ord, slot) : 255;
            else o = cmap.off_owner(ord, slot);
            s_own[i] = (unsigned char)o;
        }
        for (int gq = tid; gq < P; gq += NT) {
            const int ord = s_ord[gq];
            const bool dg = s_pa[gq] == s_pb[gq];
            s_sepown[gq] = (unsigned char)(dg ? 255 : cmap.sep_owner(ord));
            s_elneed[gq] = (dg ? cmap.diag_needed(ord, member) : cmap.off_needed(ord, member)) ? 1 : 0;
        }
        for (int a = tid; a < D; a += NT) s_meanown[a] = (unsigned char)cmap.mean_owner(a);
        __syncthreads();
        if (tid == 0) {
            // LDS slots of the pairs this member holds records of (s_elneed covers both forms of an off-diagonal pair); more than
            // the host provided cannot happen (same ClusterMap) -- checked all the same: the launch then poisons its output
            int n = 0;
            for (int gq = 0; gq < P; ++gq) s_slot[gq] = s_elneed[gq] ? n++ : -1;
            if (n > GL) *s_fail = 1;
        }
        // the element-wise slots of every pair that are this member's, as a mask (up to 64 slots per pair: the step's own-item
        // list is then a few bit operations per pair; longer pairs take the list walk)
        for (int gq = tid >> 6; gq < P; gq += NT / 64) {
            const int k = tid & 63;
            const int lim = (s_pa[gq] == s_pb[gq]) ? wtri_c : wpp;
            const unsigned long long m = __ballot(k < lim && k < wpp && s_own[gq * wpp + (k < wpp ? k : 0)] == member);
            if (k == 0) s_mymask[gq] = m;
        }
        __syncthreads();
    }
    [[maybe_unused]] auto item_owner = [&](int code) -> int {
        if (code >= 0xffff - 15) return s_meanown[0xffff - code];
        const int gq = p.magic_wpp ? (int)__umulhi((unsigned)code, p.magic_wpp) : code;
        return (s_K[gq] & 64) ? s_sepown[gq] : s_own[code];
    };
    // does this member own an item of pair gq (then it needs the pair's per-point factors)?
    [[maybe_unused]] auto pair_needed = [&](int gq) -> bool { return (s_K[gq] & 64) ? s_sepown[gq] == member : s_elneed[gq] != 0; };
    // one value of the step's exchange: two 8-byte words {tag | low}, {tag | high}
    // Two buffers, two protocols.  When every member of the candidate sits on ONE XCD -- verified below from HW_REG_XCC_ID, not
    // assumed from the dispatch order -- they share that XCD's L2: stores that stay in L2 (sc0) to ordinary device memory are visible
    // to the other members' L1-bypassing (sc1) loads after one L2 round trip.  Members on several XCDs (and the prologue that finds
    // out) use an UNCACHED buffer with write-through stores and sc1 loads: per-XCD L2s are not coherent with each other, and a clean
    // line that a reader's own L2 still holds -- from the fill kernel that zeroed the buffer, from an earlier launch -- would be
    // served to its sc1 loads for the whole bounded wait (seen in round 6 on freshly zeroed ordinary memory); uncached memory has
    // no such lines.  A different placement changes the speed, never the protocol's validity.
    [[maybe_unused]] unsigned x_tag = 0;
    [[maybe_unused]] unsigned long long* x_cur = xbuf;
    [[maybe_unused]] bool same_xcd = false;
    [[maybe_unused]] auto publish = [&](int vi, double v) {
        unsigned long long* g = x_cur + 2 * (size_t)vi;
        const unsigned long long hi_tag = (unsigned long long)x_tag << 32;
        if (same_xcd) {
            __hip_atomic_store(g, hi_tag | (unsigned)__double2loint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(g + 1, hi_tag | (unsigned)__double2hiint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            __hip_atomic_store(g, hi_tag | (unsigned)__double2loint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(g + 1, hi_tag | (unsigned)__double2hiint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    [[maybe_unused]] auto acquire = [&](int vi) -> double {
        const unsigned long long* g = x_cur + 2 * (size_t)vi;
#if defined(GPMPC_CL_DEBUG)
        if (p.cl_dbg & 1) return 0.0;                                  // timing experiments: no wait at all / one read, no check
        if (p.cl_dbg & 2) { const unsigned long long a = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return (double)a; }
#endif
        for (int spins = 0; spins < (1 << 21); ++spins) {           // bounded: a member that never arrives must not hang the GPU
            const unsigned long long a = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long b = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(a >> 32) == x_tag && (unsigned)(b >> 32) == x_tag) return __hiloint2double((int)(unsigned)b, (int)(unsigned)a);
            __builtin_amdgcn_s_sleep(1);
        }
        *s_fail = 1;
        return __builtin_nan("");
    };

    if constexpr (CL) {
        // prologue: the members' XCD ids through the placement-independent form (second buffer, tag of "step -1")
        // (its words of the ordinary buffer are only ever written by these write-through stores and by the write-through zeroing:
        //  the sc1-store / sc1-load hand-off of MI355X_MICROARCH.md, valid across XCDs; through the uncached twin it cost ~5 us more)
        x_tag = p.xch_tag0;
        x_cur = xbuf + (size_t)2 * p.xch_n;
        const int my_xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11));       // hwreg(HW_REG_XCC_ID, 0, 4)
        if (tid == 0) publish(x_xcc + member, (double)my_xcc);
        if (tid < 64) {
            const bool differs = tid < CS && acquire(x_xcc + tid) != (double)my_xcc;
            const unsigned long long any_differs = __ballot(differs);
            if (tid == 0) s_fail[1] = (any_differs == 0 && !(p.cl_dbg & 4)) ? 1 : 0;        // (cluster_debug 4: the general form anyway, A/B)
        }
        __syncthreads();
        same_xcd = s_fail[1] != 0;
        if (*s_fail) {                                 // a member never arrived (bounded wait): poison the trajectory
            if (member == 0) {
                for (int i = tid; i < H * D; i += NT) p.mu_out[((size_t)c * (H + 1) + 1) * D + i] = __builtin_nan("");
                for (int i = tid; i < H * D * D; i += NT) p.Sig_out[((size_t)c * (H + 1) + 1) * D * D + i] = __builtin_nan("");
            }
            return;
        }
    }

    int cur = 0;
    const int tid_outer = tid;
    for (int t = TILED ? p.t_begin : 0; t < (TILED ? p.t_end : H); ++t) {
        t_dbg = t;
        // Re-derive the thread coordinates inside every step from an opaque copy: otherwise the compiler
        // hoists dozens of per-thread address computations out of the horizon loop and keeps them live
        // across all phases, which overflows the 128-VGPR budget of a 1024-thread workgroup (spills).
        int tid_opaque = tid_outer;
        asm volatile("" : "+v"(tid_opaque));
        const int tid = tid_opaque;
        const int lane = tid & 63;
        const int wave = tid >> 6;
        const double* s_Sig = s_Sig2 + cur * SD2;
        double* s_SigNext = s_Sig2 + (cur ^ 1) * SD2;
        if constexpr (CL) {
            // two exchange buffers by step parity: a member writes step t + 2 only after every member published t + 1, i.e. read t
            x_tag = p.xch_tag0 + (unsigned)t + 1u;
            x_cur = (same_xcd ? xbuf : xbuf_uc) + (size_t)(t & 1) * 2 * p.xch_n;
        }

        for (int q0 = 0; q0 < P || (TILED && q0 == 0); q0 += G) {       // TILED, D = 1: no pair, the mean part still runs
            const int Gc = (P - q0 < G) ? (P - q0) : G;
            const bool first = (q0 == 0);
            const int nmean = first ? D : 0;

            // ---- P1: small D x D algebra, one thread per problem ------------------------------
            // (mean problems on wave 0, pair problems on wave 1: the two instruction streams are long dependent fp64
            //  chains and would run one after the other as divergent branches of one wavefront)
            constexpr int kPairBase = (NT >= 256) ? 64 : 0;
#if defined(GPMPC_PROF_ON)
            const long long prof_p1 = __builtin_readcyclecounter();
#endif
            if (first) {
                // input mean of this step: [mu, a_t, (time)]  (gp_model.py:98-102)
                for (int i = tid; i < E; i += NT) {
                    double v;
                    if (i < D) v = s_mu[i];
                    else if (i < DA) v = c_act[t * A + (i - D)];
                    else v = p.time0 + (double)t;
                    s_m[i] = v;
                }
            }
            if (tid < nmean) {
                // A_a = Sigma + diag(l_a^2) -> A_a^-1, det  (restated B of gp_model.py:141)
                const int a = tid;
                double* aug = s_aug + a * (D * LD);
                double prodil = 1.0;
                double detA;
                if constexpr (DP <= 4) {
                    double m[DP][2 * DP];
#pragma unroll
                    for (int i = 0; i < DP; ++i) {
                        const double il2 = (i < D) ? c_ils2[a * E + i] : 1.0;
                        prodil *= il2;
#pragma unroll
                        for (int j = 0; j < DP; ++j) {
                            const bool in = (i < D && j < D);
                            m[i][j] = (in ? s_Sig[i * D + j] : 0.0) + (i == j ? 1.0 / il2 : 0.0);
                            m[i][DP + j] = (in && i == j) ? 1.0 : 0.0;
                        }
                    }
                    detA = small_solve<DP>(m);
#pragma unroll
                    for (int i = 0; i < DP; ++i)
#pragma unroll
                        for (int j = 0; j < DP; ++j)
                            if (i < D && j < D) aug[i * LD + D + j] = m[i][DP + j];
                } else {
                    for (int i = 0; i < D; ++i) {
                        const double il2 = c_ils2[a * E + i];
                        prodil *= il2;
                        for (int j = 0; j < D; ++j) {
                            aug[i * LD + j] = s_Sig[i * D + j] + (i == j ? 1.0 / il2 : 0.0);
                            aug[i * LD + D + j] = (i == j ? 1.0 : 0.0);
                        }
                    }
                    detA = gauss_solve(aug, D, D, LD);
                }
                s_cc[a] = c_var[a] / sqrt(detA * prodil);            // c_a = var_a / sqrt(det B_a)  (:150)
            } else if (tid >= kPairBase + (kPairBase ? 0 : nmean) && tid < kPairBase + (kPairBase ? 0 : nmean) + Gc) {
                const int gq = tid - kPairBase - (kPairBase ? 0 : nmean);
                const int a = s_pa[q0 + gq], b = s_pb[q0 + gq];
                double* aug = s_aug + (D + gq) * (D * LD);
                double detR;
                double cmax = 0.0;
                // |g_i . w_j| <= sum_dd' |Z_dd'| umax_d wmax_d' with the data range of the memory points
                if constexpr (DP <= 4) {
                    double m[DP][2 * DP];
#pragma unroll
                    for (int i = 0; i < DP; ++i)
#pragma unroll
                        for (int j = 0; j < DP; ++j) {
                            const bool in = (i < D && j < D);
                            const double sg = in ? s_Sig[i * D + j] : 0.0;
                            const double dab = in ? c_ils2[a * E + j] + c_ils2[b * E + j] : 0.0;
                            m[i][j] = sg * dab + (i == j ? 1.0 : 0.0);                   // R (:156-159)
                            m[i][DP + j] = sg;
                        }
                    double ur[DP], wr[DP];                 // data-range bounds of |u_d|, |w_d|: independent of the solve
#pragma unroll
                    for (int i = 0; i < DP; ++i) {
                        const double mi = (i < D) ? s_mu[i] : 0.0;
                        const double rg = (i < D) ? fmax(fabs(c_xr[i] - mi), fabs(c_xr[E + i] - mi)) : 0.0;
                        ur[i] = (i < D) ? rg * c_ils2[a * E + i] : 0.0;
                        wr[i] = (i < D) ? rg * c_ils2[b * E + i] : 0.0;
                    }
                    detR = small_solve<DP>(m);                                            // Z = R^-1 Sigma = 2Q (:163)
                    double rowc[DP];                       // row sums first: three short chains instead of one of D^2 FMAs
#pragma unroll
                    for (int i = 0; i < DP; ++i) {
                        double r = 0.0;
#pragma unroll
                        for (int j = 0; j < DP; ++j) {
                            if (i < D && j < D) {
                                aug[i * LD + D + j] = m[i][DP + j];
                                r = fma(fabs(m[i][DP + j]), wr[j], r);
                            }
                        }
                        rowc[i] = r * ur[i];
                    }
#pragma unroll
                    for (int i = 0; i < DP; ++i) cmax += rowc[i];
                } else {
                    for (int i = 0; i < D; ++i)
                        for (int j = 0; j < D; ++j) {
                            const double dab = c_ils2[a * E + j] + c_ils2[b * E + j];
                            aug[i * LD + j] = s_Sig[i * D + j] * dab + (i == j ? 1.0 : 0.0);
                            aug[i * LD + D + j] = s_Sig[i * D + j];
                        }
                    detR = gauss_solve(aug, D, D, LD);
                    const double* Z = aug + D;
                    for (int i = 0; i < D; ++i) {
                        const double mi = s_mu[i];
                        const double ui = fmax(fabs(c_xr[i] - mi), fabs(c_xr[E + i] - mi)) * c_ils2[a * E + i];
                        for (int j = 0; j < D; ++j) {
                            const double mj = s_mu[j];
                            const double wj = fmax(fabs(c_xr[j] - mj), fabs(c_xr[E + j] - mj)) * c_ils2[b * E + j];
                            cmax = fma(fabs(Z[i * LD + j]) * ui, wj, cmax);
                        }
                    }
                }
                s_rdet[gq] = 1.0 / sqrt(detR);                                           // (:176)
                // smallest K with cmax^(K+1)/(K+1)! * exp(2 cmax) <= 2^-54 (truncation below fp64 rounding)
                int K = 0;
                if (p.force_path != 1 && cmax <= kTaylorMaxArg[kMaxTaylor]) {
                    K = 1;
#pragma unroll
                    for (int k = 1; k < kMaxTaylor; ++k) K += (cmax > kTaylorMaxArg[k]) ? 1 : 0;
                }
                // separable (moment) evaluation of an off-diagonal pair when it is the cheaper one
                if (a != b && K > 0 && K <= p.sep_kmax && (DX != 3 || K <= 6) && p.force_path == 0) {
                    // wavefront instructions against ~(D + K + 3) per element of the pairwise loop.  D = 3 (compile-time
                    // monomial structure): ~2 per monomial and 64 points + the 8-value reductions; otherwise per block of 8
                    // monomials and side ~(6 + 8 (avg degree + 1)) per 64 points + one 8-value reduction
                    const long long C_ = s_mcum[K < 3 ? 3 : K];
                    const long long NBk = (s_mcum[K] + 7) / 8;
                    const long long cost_sep = (DX == 3) ? 2 * (((N + 63) / 64) * (2 * C_ + 40) + (C_ / 8 + 4) * 70)
                                                         : 2 * NBk * (((N + 63) / 64) * (6 + 8 * K) + 80);
                    const long long cost_el = (long long)N * N * (D + K + 3) / 64;
                    if (p.force_sep || cost_sep < cost_el) K |= 64;
                }
                // (cooperative form: the LDS slot of the pair's records rides in the bits above the degree and the separable flag, so the
                //  readers of s_K get it without another dependent LDS read)
                if constexpr (CL) K |= (s_slot[gq] + 1) << 8;
                s_K[gq] = K;
            }
            if constexpr (TILED) {
                if (first && tid >= 128 && tid < 128 + D) {
                    // diagonal pair (a, a): only det R is needed here (Z and the degree: tile_params_kernel)
                    const int a = tid - 128;
                    double m[DP][2 * DP];
#pragma unroll
                    for (int i = 0; i < DP; ++i)
#pragma unroll
                        for (int j = 0; j < DP; ++j) {
                            const bool in = (i < D && j < D);
                            const double sg = in ? s_Sig[i * D + j] : 0.0;
                            const double dab = in ? c_ils2[a * E + j] + c_ils2[a * E + j] : 0.0;
                            m[i][j] = sg * dab + (i == j ? 1.0 : 0.0);
                            m[i][DP + j] = sg;
                        }
                    s_rdiag[a] = 1.0 / sqrt(small_solve<DP>(m));
                } else if (first && wave == 3) {
                    // sums of the per-tile partial sums of pair_tile_kernel, fixed order
                    for (int a = 0; a < D; ++a) {
                        const double* tp = p.tile_part + ((size_t)c * D + a) * p.ntiles;
                        double v = 0.0;
                        for (int k = lane; k < p.ntiles; k += 64) v += tp[k];
                        v = wave_sum(v);
                        if (lane == 0) s_rdiag[4 + a] = v;
                    }
                }
            }
#if defined(GPMPC_PROF_ON)
            if (threadIdx.x == 0 && blockIdx.x == 0) { long long now_ = __builtin_readcyclecounter(); prof_acc[7] += now_ - prof_last; }
            if (blockIdx.x == 0 && threadIdx.x == 64) prof_x[0] += __builtin_readcyclecounter() - prof_p1;     // pair problems done
#endif
            if (tid == NT - 1) {
                *s_counter = 0;
                int no = 0;
                for (int gq = 0; gq < Gc; ++gq)
                    if (s_pa[q0 + gq] != s_pb[q0 + gq]) s_off[no++] = gq;
                *s_noff = no;
#if defined(GPMPC_PROF_ON)
                if (blockIdx.x == 0) prof_x[1] += __builtin_readcyclecounter() - prof_p1;                        // last thread's bookkeeping done
#endif
            }
            __syncthreads();
            GPMPC_TRACE(2);

            // ---- P2: per-point quantities ------------------------------------------------------
            // Three kinds of items of about one exp each, so that the passes over the 1024 threads stay balanced:
            // mean part (output a), row side of a pair (u, g = Z^T u, ka'), column side of an off-diagonal pair (w, kb');
            // for a diagonal pair the column factor is the row factor.
            const int n_off = *s_noff;
            [[maybe_unused]] bool cl_list_pending = false;
            [[maybe_unused]] int cl_ns = 0, cl_cls = 3, cl_code0 = 0;
            if (wave == NW - 1) {
                // Work-item list of this group (the Taylor degrees / evaluation forms of P1 are visible after the barrier), built
                // by the lanes of ONE wavefront in parallel: lane l < Gc owns pair l, the next nmean lanes a mean-sum item each
                // (Gc + nmean <= 64: the host caps G).  Classes in list order: element-wise pairs, separable pairs, mean sums.
                // (A first version built the list on one thread: ~40 dependent LDS round trips, 4.3 k cycles per step.)
                const int wtri = (s_tri[p.RC] + 63) >> 6;
                const int R = Gc + nmean;
                int ns = 0, cls = 3, code0 = 0;
                [[maybe_unused]] unsigned long long mymask = 0;           // cooperative form: this entry's items that are this member's
                if (lane < Gc) {
                    const int Kr = s_K[lane];
                    const bool sep = (Kr & 64) != 0;
                    if (sep) {
                        const int Ks = Kr & 63;
                        const int two = (DX == 3) ? 2 * sep3_bands(Ks <= 3 ? 3 : Ks) : 2 * ((s_mcum[Ks] + 7) >> 3);
                        ns = two < wpp ? two : wpp;
                    } else {
                        ns = (s_pa[q0 + lane] == s_pb[q0 + lane]) ? wtri : wpp;
                    }
                    cls = sep ? 1 : 0;
                    code0 = lane * wpp;
                    s_nslot[lane] = ns;
                    if constexpr (CL) {
                        if (sep) mymask = (s_sepown[lane] == member) ? (ns >= 64 ? ~0ull : (1ull << ns) - 1ull) : 0ull;
                        else mymask = s_mymask[lane];
                    }
                } else if (lane < R) {
                    ns = 1; cls = 2; code0 = 0xffff - (lane - Gc);
                    if constexpr (CL) mymask = (s_meanown[lane - Gc] == member) ? 1ull : 0ull;
                }
                // the full list (class order): start = items of all entries that come before this lane's in (class, lane) order
                auto build_list = [&] {
                    int start = 0, total_items = 0;
                    for (int m = 0; m < R; ++m) {
                        const int nsm = __builtin_amdgcn_readlane(ns, m);
                        const int clm = __builtin_amdgcn_readlane(cls, m);
                        start += (clm < cls || (clm == cls && m < lane)) ? nsm : 0;
                        total_items += nsm;
                    }
                    for (int k = 0; k < ns; ++k) s_items[start + k] = (unsigned short)(cls == 2 ? code0 : code0 + k);
                    if (lane == 0) *s_nitems = total_items;
                    return total_items;
                };
                if constexpr (!CL) {
                    (void)build_list();
                } else if (wpp > 64) {
                    // (pairs of more than 64 slots: the list first, then this member's items by walking it)
                    const int total_items = build_list();
                    wave_lds_sync();
                    int cnt = 0;
                    for (int base = 0; base < total_items; base += 64) {
                        const int k = base + lane;
                        const int code = s_items[k < total_items ? k : 0];
                        const bool mine = k < total_items && item_owner(code) == member;
                        const unsigned long long mask = __ballot(mine);
                        if (mine) s_mine[1 + cnt + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned short)code;
                        cnt += __popcll(mask);
                    }
                    if (lane == 0) s_mine[0] = (unsigned short)cnt;
                } else {
                    // Cooperative form: what P3 needs right behind the barrier is THIS member's items (no queue: the j-th goes to
                    // wavefront j mod NW; in the list's class order -- the long element-wise items first, then the separable pairs',
                    // then the mean sums) -- a few bit operations per entry.  The full list
                    // (read by the gather at the END of P3, and its length) is built behind the barrier, beside the items: built
                    // here first it made this wavefront the last one at the barrier by ~2 k cycles per step (round 6: 5.6 k of
                    // this lone wavefront's dependent instructions against 2-4.7 k of the per-point pass).
                    const int mycnt = __popcll(mymask);
                    int pre = 0, tot = 0, total_items = 0;
                    for (int m = 0; m < R; ++m) {
                        const int cm = __builtin_amdgcn_readlane(mycnt, m);
                        const int clm = __builtin_amdgcn_readlane(cls, m);
                        pre += (clm < cls || (clm == cls && m < lane)) ? cm : 0;
                        tot += cm;
                        total_items += __builtin_amdgcn_readlane(ns, m);
                    }
                    unsigned long long mk = mymask;
                    for (int i = 0; mk != 0; ++i) {
                        const int k = __builtin_ctzll(mk);
                        mk &= mk - 1ull;
                        s_mine[1 + pre + i] = (unsigned short)(cls == 2 ? code0 : code0 + k);
                    }
                    if (lane == 0) { s_mine[0] = (unsigned short)tot; *s_nitems = total_items; }
                    cl_list_pending = true;
                    cl_ns = ns; cl_cls = cls; cl_code0 = code0;
                }
            }
            // The per-point items go to all wavefronts but the last one, which builds the work-item list above beside them (with
            // items of its own it was the last to reach the barrier: + 1.9 k cycles per step at config 2); narrow workgroups
            // (fewer than 8 wavefronts) share the items among all.
            const int p2_threads = (NW >= 8) ? NT - 64 : NT;
            // (Measured and not adopted, round 5: two pair-side items per thread and trip, and two pair-side + one mean item,
            //  written stage by stage so that the dependent chains interleave -- no gain for two (0.427 vs 0.428 ms at config 2),
            //  slower for three (0.400 -> 0.415 ms; pass 6.5 k -> 7.9 k cycles per step; config 1 -10 %, B = 4096 per GPU -38 % with
            //  the spills of the wider live set): the pass is not bound by the latency of one thread's chain.
            //  profiles/r05c_forward_ab.txt, profiles/r05e_p2_three_chains_*.txt)
            int p2_items = (nmean + Gc + n_off) * N;
            [[maybe_unused]] bool cl_nu_first = false;
            if (CL && (NW < 8 || wave != NW - 1)) {       // (the list-building wavefront of a wide member has no per-point items)
                // only the problems this member owns items of, compacted per wavefront (lane = problem; 255 = nu alone, when
                // the mean problem of output 0, which stores nu, is another member's)
                const int nall = nmean + Gc + n_off;
                bool need = false;
                int code = lane;
                if (lane < nmean) need = s_meanown[lane] == member;
                else if (lane < nmean + Gc) need = pair_needed(lane - nmean);
                else if (lane < nall) need = pair_needed(s_off[lane - nmean - Gc]);
                // (nu = X - m, which the mean problem of output 0 stores, is stored by the member's FIRST problem otherwise -- every
                //  problem visits all points -- or, for a member without problems, by the problem 255 that does nothing else)
                const unsigned long long mask0 = __ballot(need);
                if (lane == nall) { need = s_meanown[0] != member && mask0 == 0; code = 255; }
                const unsigned long long mask = __ballot(need);
                if (need) s_needw[wave * 64 + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned char)code;
                p2_items = __popcll(mask) * N;
                cl_nu_first = s_meanown[0] != member && mask0 != 0;
                wave_lds_sync();
            }
            for (int it = tid; it < p2_items && tid < p2_threads; it += p2_threads) {
                int prob = p.magic_pt ? (int)__umulhi((unsigned)it, p.magic_pt) : it;      // it / N
                const int pt = it - prob * N;
                double nu[DP];
#pragma unroll
                for (int d = 0; d < DP; ++d) nu[d] = (d < D) ? (Xs[d * N + pt] - s_m[d]) : 0.0;
                if constexpr (CL) {
                    if (cl_nu_first && prob == 0) {
#pragma unroll
                        for (int d = 0; d < DP; ++d)
                            if (d < D) a_nu[d * N + pt] = nu[d];
                    }
                    prob = s_needw[wave * 64 + prob];
                    if (prob == 255) {
#pragma unroll
                        for (int d = 0; d < DP; ++d)
                            if (d < D) a_nu[d * N + pt] = nu[d];
                        continue;
                    }
                }
                if (prob < nmean) {
                    const int a = prob;
                    const double* Ai = s_aug + a * (D * LD) + D;          // A_a^-1
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
                        const double v = Xs[e * N + pt] - s_m[e];
                        q = fma(v * v, c_ils2[a * E + e], q);
                    }
                    a_lb[a * N + pt] = fast_exp(-0.5 * q, c_exptab) * p.beta[a * N + pt];   // lb (:148)
                    if (a == 0) {
#pragma unroll
                        for (int d = 0; d < DP; ++d)
                            if (d < D) a_nu[d * N + pt] = nu[d];
                    }
                } else {
                    const bool rowside = prob < nmean + Gc;
                    const int gq = rowside ? prob - nmean : s_off[prob - nmean - Gc];
                    const int a = s_pa[q0 + gq], b = s_pb[q0 + gq];
                    const int c = rowside ? a : b;                         // the output whose lengthscales scale nu
                    const int Kw = s_K[gq];
                    const int K = Kw & 63;
                    const int sl = CL ? (Kw >> 8) - 1 : gq;                // LDS slot of the pair's records
                    const double* Z = s_aug + (D + gq) * (D * LD) + D;
                    double u[DP], g[DP];
                    double ks = 0.0;                                       // sum_e nu_e^2 / l_e^2
#pragma unroll
                    for (int d = 0; d < DP; ++d) {
                        u[d] = nu[d] * ((d < D) ? c_ils2[c * E + d] : 0.0);
                        ks = fma(nu[d], u[d], ks);
                        g[d] = 0.0;
                    }
                    for (int e = D; e < E; ++e) {
                        const double v = Xs[e * N + pt] - s_m[e];
                        ks = fma(v * v, c_ils2[c * E + e], ks);
                    }
                    double qq = 0.0;
#pragma unroll
                    for (int i = 0; i < DP; ++i) {
                        if (i < D) {
                            double zu = 0.0;
#pragma unroll
                            for (int j = 0; j < DP; ++j)
                                if (j < D) {
                                    const double z = Z[i * LD + j];
                                    zu = fma(z, u[j], zu);
                                    g[j] = fma(z, u[i], g[j]);      // g = Z^T u: cross term u^T Z w = g . w
                                }
                            qq = fma(u[i], zu, qq);
                        }
                    }
                    const double kk = c_logvar[c] - 0.5 * ks + 0.5 * qq;                  // k (:168) + u^T Q u
                    const double bc = p.beta[c * N + pt];
                    if (rowside) {
                        double* rec = a_rows + ((size_t)sl * NR + pt) * RS;
#pragma unroll
                        for (int d = 0; d < DP; ++d) rec[REC::G + d] = g[d];
                        if (K > 0) {
                            const double ea = fast_exp(kk, c_exptab);
                            rec[REC::EA] = ea;
                            rec[REC::RA] = ea * bc;
                            if (a == b) a_kb[sl * N + pt] = ea;
                        } else {
                            rec[REC::EA] = kk;
                            rec[REC::RA] = bc;
                            if (a == b) a_kb[sl * N + pt] = kk;
                        }
                        if constexpr (REC::RS > DP + 2) rec[DP + 2] = 0.0;          // the pad travels with the 16-byte reads
                    } else {
                        a_kb[sl * N + pt] = (K > 0) ? fast_exp(kk, c_exptab) * bc : kk;
                    }
                }
            }
            __syncthreads();
            GPMPC_TRACE(3);

            // ---- P3: work queue: pairwise items, moment sums, mean sums, stage cost ----------------
            const int total = *s_nitems;
            if constexpr (CL) {
                if (cl_list_pending) {           // the full list, for the gather behind the items (a barrier in between)
                    const int R = Gc + nmean;
                    int start = 0;
                    for (int m = 0; m < R; ++m) {
                        const int nsm = __builtin_amdgcn_readlane(cl_ns, m);
                        const int clm = __builtin_amdgcn_readlane(cl_cls, m);
                        start += (clm < cl_cls || (clm == cl_cls && m < lane)) ? nsm : 0;
                    }
                    // entry by entry, a lane per slot (<= 64 slots per entry on this path): a lane per entry walking its slots was
                    // up to 63 dependent trips
                    for (int m = 0; m < R; ++m) {
                        const int nsm = __builtin_amdgcn_readlane(cl_ns, m), sm = __builtin_amdgcn_readlane(start, m);
                        const int clm = __builtin_amdgcn_readlane(cl_cls, m), c0 = __builtin_amdgcn_readlane(cl_code0, m);
                        if (lane < nsm) s_items[sm + lane] = (unsigned short)(clm == 2 ? c0 : c0 + lane);
                    }
                }
            }
#if defined(GPMPC_PROF_ON)
            const long long prof_p3 = __builtin_readcyclecounter();
#endif
            auto pull_item = [&]() -> int {
                int pulled = 0;
                if (lane == 0) {
                    pulled = __hip_atomic_fetch_add(s_counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    pulled = (pulled < total) ? (int)s_items[pulled] : -1;
                }
                return __builtin_amdgcn_readfirstlane(pulled);       // wave-uniform (SGPR) work item
            };
            // cooperative form: no queue -- the j-th item this member owns (listed beside the per-point pass) goes to wavefront j mod NW
            [[maybe_unused]] int cl_next = wave;
            [[maybe_unused]] const int cl_count = CL ? (int)s_mine[0] : 0;
            auto next_item = [&]() -> int {
                if constexpr (!CL) return pull_item();
                else {
                    if (cl_next >= cl_count) return -1;
                    const int code = __builtin_amdgcn_readfirstlane((int)s_mine[1 + cl_next]);
                    cl_next += NW;
                    return code;
                }
            };
            for (int wq = next_item(); wq >= 0; wq = next_item()) {
                if (wq >= 0xffff - 15) {
                    // s1[a][0] = sum_p lb, s1[a][1+d] = sum_p lb nu_d  (fixed order)
                    const int a = 0xffff - wq;
                    for (int dd = 0; dd <= D; ++dd) {
                        double v = 0.0;
                        if (dd == 0) { for (int pt = lane; pt < N; pt += 64) v += a_lb[a * N + pt]; }
                        else { for (int pt = lane; pt < N; pt += 64) v = fma(a_lb[a * N + pt], a_nu[(dd - 1) * N + pt], v); }
                        v = wave_sum(v);
                        if (lane == 0) {
                            s_s1[a * (D + 1) + dd] = v;
                            if constexpr (CL) publish(x_mean + a * (D + 1) + dd, v);
                        }
                    }
                    continue;
                }
                const int wi = wq;
                const int gq = p.magic_wpp ? (int)__umulhi((unsigned)wi, p.magic_wpp) : wi;   // wi / wpp (magic 0: wpp == 1)
                const int slot = wi - gq * wpp;
                const int a = __builtin_amdgcn_readfirstlane(s_pa[q0 + gq]);
                const int b = __builtin_amdgcn_readfirstlane(s_pb[q0 + gq]);
                const int Kraw = __builtin_amdgcn_readfirstlane(s_K[gq]);
                const int K = Kraw & 63;
                const bool diag = (a == b);
                if (Kraw & 64) {
                    // separable evaluation: moments  G_alpha = sum_i ra_i g_i^alpha,  W_alpha = sum_j rb_j w_j^alpha.
                    // Item = (side, block of 8 consecutive monomials): lanes own points, the point's x and weight are loaded
                    // once for the 8 monomials, whose exponents are wave-uniform (scalar loop counts).
                    [[maybe_unused]] const int sl3 = CL ? (Kraw >> 8) - 1 : gq;
                    if constexpr (DX == 3) {
                        // three state dimensions: compile-time monomial structure, items = (side, band of the x0 exponent)
                        const int KS = K <= 3 ? 3 : K;                                   // K <= 6 on this path (P1)
                        const int nbands = sep3_bands(KS);
                        for (int itm = slot; itm < 2 * nbands; itm += wpp) {
                            const int side = itm >= nbands, band = itm - side * nbands;
                            const double* rec0 = a_rows + (size_t)sl3 * NR * RS;
                            const double* kbp = a_kb + sl3 * N;
                            const double* il = c_ils2 + b * E;
                            double* mom = s_mom + (gq * 2 + side) * rnd2(CM);
                            if (K <= 3) sep3_item<3>(band, lane, N, side, rec0, RS, kbp, a_nu, il, mom);
                            else if (K == 4) sep3_item<4>(band, lane, N, side, rec0, RS, kbp, a_nu, il, mom);
                            else if (K == 5) sep3_item<5>(band, lane, N, side, rec0, RS, kbp, a_nu, il, mom);
                            else sep3_item<6>(band, lane, N, side, rec0, RS, kbp, a_nu, il, mom);      // K <= 6 on this path (P1)
                        }
                        continue;
                    }
                    const int C = s_mcum[K];
                    const int NBk = (C + 7) >> 3;
                    for (int blk = slot; blk < 2 * NBk; blk += wpp) {
                        const int side = blk >= NBk;
                        const int al0 = (blk - side * NBk) * 8;
                        int packed[8];
#pragma unroll
                        for (int m = 0; m < 8; ++m) packed[m] = __builtin_amdgcn_readfirstlane((al0 + m < C) ? c_monoe[al0 + m] : 0);
                        double acc[8];
#pragma unroll
                        for (int m = 0; m < 8; ++m) acc[m] = 0.0;
                        for (int pt = lane; pt < N; pt += 64) {
                            double x[DP];
                            double wt;
                            if (side == 0) {
                                double r[RS];
                                REC::template load<RS>(a_rows + ((size_t)sl3 * NR + pt) * RS, r);
                                wt = r[REC::RA];
#pragma unroll
                                for (int d = 0; d < DP; ++d) x[d] = r[REC::G + d];
                            } else {
                                wt = a_kb[sl3 * N + pt];
#pragma unroll
                                for (int d = 0; d < DP; ++d) x[d] = (d < D) ? a_nu[d * N + pt] * c_ils2[b * E + d] : 0.0;
                            }
#pragma unroll
                            for (int m = 0; m < 8; ++m) {
                                double t = wt;
#pragma unroll
                                for (int d = 0; d < (DP < 4 ? DP : 4); ++d) {
                                    const int ne = (packed[m] >> (8 * d)) & 255;
                                    for (int e = 0; e < ne; ++e) t *= x[d];
                                }
                                acc[m] += t;
                            }
                        }
                        const double tot = wave_reduce8(acc);          // lanes 4 m .. 4 m + 3 (m < 8) hold the total of value m
                        const int mm = lane >> 2;
                        if ((lane & 3) == 0 && lane < 32 && al0 + mm < C) s_mom[(gq * 2 + side) * rnd2(CM) + al0 + mm] = tot;
                    }
                    continue;
                }
                const int flat = slot * 64 + lane;
                bool valid;
                int r, jc;
                int nrows_tab = -1;
                if (diag && use_lmap) {
                    const int packed = s_lmap[flat];
                    valid = packed >= 0;
                    r = valid ? (packed & 255) : 0;
                    jc = valid ? (packed >> 8) : 0;
                    nrows_tab = __builtin_amdgcn_readfirstlane(s_lrows[slot]);
                } else if (diag) {
                    // only the (row chunk, column unit) combinations that contain an element i <= j are enumerated:
                    // s_tri[r] = number of such combinations in chunks < r (lanes binary-search their chunk)
                    valid = flat < s_tri[p.RC];
                    int lo = 0, hi = p.RC;                         // invariant: s_tri[lo] <= flat < s_tri[hi]
                    while (hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if (s_tri[mid] <= flat) lo = mid; else hi = mid;
                    }
                    r = valid ? lo : 0;
                    const int first = C2 ? (r * p.CH) / 2 : r * p.CH;
                    jc = valid ? first + (flat - s_tri[r]) : 0;
                } else {
                    valid = flat < p.RC * NC;
                    r = valid ? (p.magic_N ? (int)__umulhi((unsigned)flat, p.magic_N) : flat) : 0;   // flat / NC
                    jc = valid ? flat - r * NC : 0;
                }
                const int j = C2 ? 2 * jc : jc;
                const bool valid1 = C2 && (j + 1 < N);        // second column of the lane (two-column form)
                const int jl = valid1 ? j + 1 : j;            // last column of the lane
                const int i0 = r * p.CH;                      // CH is a multiple of 4
                int i1 = i0 + p.CH;
                if (i1 > N) i1 = N;
                if (diag && i1 > jl + 1) i1 = jl + 1;
                int nrows = nrows_tab;                             // wave-uniform, zero padding absorbs the overshoot
                if (nrows_tab < 0) {
                    const int len = (valid && i1 > i0) ? (i1 - i0) : 0;
                    nrows = (wave_max_i32(len) + 3) & ~3;
                }
                double acc = 0.0;
                if (nrows > 0) {
                    double w[DP];
#pragma unroll
                    for (int d = 0; d < DP; ++d) w[d] = (d < D) ? a_nu[d * N + j] * c_ils2[b * E + d] : 0.0;
                    const int sle = CL ? (Kraw >> 8) - 1 : gq;
                    const double kbj = a_kb[sle * N + j];
                    const double* rec = a_rows + ((size_t)sle * NR + i0) * RS;
                    const double* Tp = p.Tm + ((size_t)a * (N + kTPad) + i0) * N + j;
                    if constexpr (C2) {
                        // second column: always read (jl is a valid column) and masked by a factor -- four `valid1 ? load : 0`
                        // became four exec-mask branches with a wait each in every item's set-up
                        double w1[DP];
                        const double m1 = valid1 ? 1.0 : 0.0;
#pragma unroll
                        for (int d = 0; d < DP; ++d) w1[d] = (d < D) ? a_nu[d * N + jl] * (c_ils2[b * E + d] * m1) : 0.0;
                        const double kb1 = a_kb[sle * N + jl] * m1;
                        double acc0, acc1;
                        if (K == 0) {
                            item_exp2<DP>(rec, nrows, w, w1, kbj, kb1, diag, Tp, N, c_exptab, acc0, acc1);
                            acc = acc0 * (diag ? 2.0 : p.beta[b * N + j]) + (valid1 ? acc1 * (diag ? 2.0 : p.beta[b * N + j + 1]) : 0.0);
                        } else {
                            // rows per trip = depth of the T prefetch (see item_taylor2): 4 from DP = 4 up.  One
                            // instantiation per degree: a second one (choice by N) cost config 2 2.7 % through the larger kernel.
                            constexpr int TU = DP >= 4 ? 4 : 2;       // (the cooperative form measured the same with 2, 4, 8: its items are bound by their set-up)
                            if (K <= 2) item_taylor2<DP, 2, TU>(rec, nrows, w, w1, diag, Tp, N, acc0, acc1);
                            else if (K == 3) item_taylor2<DP, 3, TU>(rec, nrows, w, w1, diag, Tp, N, acc0, acc1);
                            else if (K == 4) item_taylor2<DP, 4, TU>(rec, nrows, w, w1, diag, Tp, N, acc0, acc1);
                            else if (K == 5) item_taylor2<DP, 5, TU>(rec, nrows, w, w1, diag, Tp, N, acc0, acc1);
                            else if (K == 6) item_taylor2<DP, 6, TU>(rec, nrows, w, w1, diag, Tp, N, acc0, acc1);
                            else if (K == 7) item_taylor2<DP, 7, TU>(rec, nrows, w, w1, diag, Tp, N, acc0, acc1);
                            else if (K == 8) item_taylor2<DP, 8, TU>(rec, nrows, w, w1, diag, Tp, N, acc0, acc1);
                            else if (K <= 10) item_taylor2<DP, 10, TU>(rec, nrows, w, w1, diag, Tp, N, acc0, acc1);
                            else if (K <= 12) item_taylor2<DP, 12, TU>(rec, nrows, w, w1, diag, Tp, N, acc0, acc1);
                            else item_taylor2<DP, 14, TU>(rec, nrows, w, w1, diag, Tp, N, acc0, acc1);
                            acc = fma(acc0, kbj, acc1 * kb1) * (diag ? 2.0 : 1.0);
                        }
                    } else if (K == 0) {
                        acc = item_exp<DP, REC>(rec, nrows, w, kbj, diag, Tp, N, c_exptab);
                        acc *= diag ? 2.0 : p.beta[b * N + j];
                    } else {
                        if (K <= 2) acc = item_taylor<DP, 2, REC>(rec, nrows, w, diag, Tp, N);
                        else if (K == 3) acc = item_taylor<DP, 3, REC>(rec, nrows, w, diag, Tp, N);
                        else if (K == 4) acc = item_taylor<DP, 4, REC>(rec, nrows, w, diag, Tp, N);
                        else if (K == 5) acc = item_taylor<DP, 5, REC>(rec, nrows, w, diag, Tp, N);
                        else if (K == 6) acc = item_taylor<DP, 6, REC>(rec, nrows, w, diag, Tp, N);
                        else if (K == 7) acc = item_taylor<DP, 7, REC>(rec, nrows, w, diag, Tp, N);
                        else if (K == 8) acc = item_taylor<DP, 8, REC>(rec, nrows, w, diag, Tp, N);
                        else if (K <= 10) acc = item_taylor<DP, 10, REC>(rec, nrows, w, diag, Tp, N);
                        else if (K <= 12) acc = item_taylor<DP, 12, REC>(rec, nrows, w, diag, Tp, N);
                        else acc = item_taylor<DP, 14, REC>(rec, nrows, w, diag, Tp, N);
                        acc *= diag ? 2.0 * kbj : kbj;
                    }
                    acc = valid ? acc : 0.0;
                }
                acc = wave_sum(acc);
                if (lane == 0) {
                    s_part[wi] = acc;
                    if constexpr (CL) publish(wi, acc);
                }
            }
#if defined(GPMPC_PROF_ON) && defined(GPMPC_PROF_BARRIER)
            if constexpr (CL) __syncthreads();           // prof A/B: "items" then ends when the member's LAST wavefront is done
#endif
            GPMPC_TRACE(8);
            if constexpr (CL) {
                // totals of the separable pairs this member owns (its wavefronts filled the pair's moments), then every value of the
                // step from whichever member formed it
                const bool own_sep = __ballot(lane < Gc && (s_K[lane < Gc ? lane : 0] & 64) && s_sepown[lane < Gc ? lane : 0] == member) != 0;
                __syncthreads();                 // (the moments of the separable pairs this member owns; the list built behind the P2 barrier)
                if (own_sep) {
                    for (int gq = wave; gq < Gc; gq += NW) {
                        const int Kraw = s_K[gq];
                        if (!(Kraw & 64) || s_sepown[gq] != member) continue;
                        const int C = s_mcum[Kraw & 63];
                        const double* Gm = s_mom + (gq * 2) * rnd2(CM);
                        const double* Wm = Gm + rnd2(CM);
                        double v = 0.0;
                        for (int al = lane; al < C; al += 64) v = fma(Gm[al] * Wm[al], c_monow[al], v);
                        v = wave_sum(v);
                        if (lane == 0) publish(x_sep + gq, v);
                    }
                }
                // one value per lane: the list's pair items (the mean sums are its last nmean entries), then the nmean (D + 1) mean
                // sums -- every lane runs the SAME wait loop once (divergent classes would queue their round trips one after another)
                const int npair = total - nmean;
                for (int k = tid; k < npair + nmean * (D + 1); k += NT) {
                    int vi = -1;
                    double* dst = nullptr;
                    if (k < npair) {
                        const int code = s_items[k];
                        const int gq = p.magic_wpp ? (int)__umulhi((unsigned)code, p.magic_wpp) : code;
                        if (!(s_K[gq] & 64)) { vi = code; dst = s_part + code; }
                        else if (code == gq * wpp) { vi = x_sep + gq; dst = s_sepv + gq; }
                    } else {
                        vi = x_mean + (k - npair); dst = s_s1 + (k - npair);
                    }
                    if (vi >= 0) *dst = acquire(vi);
                }
            }
#if defined(GPMPC_PROF_ON)
            if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) prof_x[2] += __builtin_readcyclecounter() - prof_p3;     // this wavefront left the queue
#endif
            __syncthreads();
            GPMPC_TRACE(4);

            // ---- P4: per-pair totals (one wave per pair, fixed order), M and V -------------------
            for (int gq = wave; gq < Gc; gq += NW) {
                const int Kraw = s_K[gq];
                double v = 0.0;
                if ((Kraw & 64) && CL) {
                    v = s_sepv[gq];                          // summed by the owning member in this very order
                } else {
                    if (Kraw & 64) {
                        const int C = s_mcum[Kraw & 63];
                        const double* Gm = s_mom + (gq * 2) * rnd2(CM);
                        const double* Wm = Gm + rnd2(CM);
                        for (int al = lane; al < C; al += 64) v = fma(Gm[al] * Wm[al], c_monow[al], v);
                    } else {
                        const int ns = s_nslot[gq];              // the slots the list held (the others were empty: exact zeros before)
                        for (int k = lane; k < ns; k += 64) v += s_part[gq * wpp + k];
                    }
                    v = wave_sum(v);
                }
                if constexpr (TILED) {
                    const int a = s_pa[q0 + gq], b = s_pb[q0 + gq];
                    if (lane == 0) s_Sp[a * D - (a * (a - 1)) / 2 + (b - a)] = v * s_rdet[gq];
                } else {
                    if (lane == 0) s_Sp[q0 + gq] = v * s_rdet[gq];
                }
            }
            if constexpr (TILED) {
                // i <= j only: factor 2
                if (first && tid < D) s_Sp[tid * D - (tid * (tid - 1)) / 2] = 2.0 * s_rdiag[4 + tid] * s_rdiag[tid];
            }
            if (first) {
                if (tid < D) s_M[tid] = s_cc[tid] * s_s1[tid * (D + 1)];                 // M_a (:152)
                for (int idx = tid; idx < D * D; idx += NT) {
                    const int k = idx / D, a = idx - k * D;
                    const double* Ai = s_aug + a * (D * LD) + D;
                    double s = 0.0;
                    for (int j = 0; j < D; ++j) s = fma(Ai[k * LD + j], s_s1[a * (D + 1) + 1 + j], s);
                    s_Vs[idx] = s_cc[a] * s;                                             // state rows of V (:153)
                }
            }
            __syncthreads();
            GPMPC_TRACE(5);
        }

        // ---- P5: state update  (gp_model.py:105-108, 177-178) ---------------------------------
        for (int idx = tid; idx < D * D; idx += NT) {
            const int i = idx / D, j = idx - i * D;
            const int a = i < j ? i : j, b = i < j ? j : i;
            const int q = a * D - (a * (a - 1)) / 2 + (b - a);
            const double S = s_Sp[q] - s_M[i] * s_M[j] + (i == j ? c_var[i] : 0.0);
            double cij = 0.0, cji = 0.0;
            for (int k = 0; k < D; ++k) {
                cij = fma(s_Sig[i * D + k], s_Vs[k * D + j], cij);
                cji = fma(s_Sig[j * D + k], s_Vs[k * D + i], cji);
            }
            const double v = S + s_Sig[idx] + (cij + cji);   // (cij + cji) commutes: Sigma stays exactly symmetric
            s_SigNext[idx] = v;
            if (member == 0) p.Sig_out[((size_t)c * (H + 1) + (t + 1)) * D * D + idx] = v;
        }
        for (int i = tid; i < D; i += NT) {
            const double v = s_mu[i] + s_M[i];
            s_mu[i] = v;
            if (member == 0) p.mu_out[((size_t)c * (H + 1) + (t + 1)) * D + i] = v;
        }
        cur ^= 1;
        __syncthreads();
        if constexpr (CL) {
            if (*s_fail) {
                // a member never arrived (bounded wait): poison the rest of the trajectory instead of spinning on
                if (member == 0) {
                    for (int i = tid; i < (H - t) * D; i += NT) p.mu_out[((size_t)c * (H + 1) + (t + 1)) * D + i] = __builtin_nan("");
                    for (int i = tid; i < (H - t) * D * D; i += NT) p.Sig_out[((size_t)c * (H + 1) + (t + 1)) * D * D + i] = __builtin_nan("");
                }
                return;
            }
        }
        GPMPC_TRACE(6);
    }
#if defined(GPMPC_PROF_ON)
    if (threadIdx.x == 0 && blockIdx.x == 0) printf("PROF wave0 small algebra cycles %lld\n", prof_acc[7]);
    if (blockIdx.x == 0 && threadIdx.x == 64) printf("PROF P1 pair problems done after %lld cycles (summed over steps)\n", prof_x[0]);
    if (blockIdx.x == 0 && threadIdx.x == NT - 1) printf("PROF P1 last thread done after %lld\n", prof_x[1]);
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) printf("PROF P3 wave %d left the queue after %lld\n", (int)(threadIdx.x >> 6), prof_x[2]);
    if (CL && threadIdx.x == 0 && (blockIdx.x & 7) == 0)
        printf("PROF member %2d xcc %d same_xcd %d\n", member, (int)__builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)), (int)same_xcd);
    if (CL && threadIdx.x == 0 && (blockIdx.x & 7) == 0)
        printf("PROF member %2d wall (10 ns ticks): P1 %lld | P2 %lld | items(wave 0) %lld | gather+wait %lld | P4 %lld | P5 %lld\n", member,
               prof_wall[2], prof_wall[3], prof_wall[8], prof_wall[4], prof_wall[5], prof_wall[6]);
    if (threadIdx.x == 0 && blockIdx.x == 0)
        printf("PROF cycles: init %lld | P1 %lld | P2 %lld | P3 %lld (wave 0's items %lld) | P4 %lld | P5 %lld\n", prof_acc[1], prof_acc[2],
               prof_acc[3], prof_acc[4] + prof_acc[8], prof_acc[8], prof_acc[5], prof_acc[6]);
#endif
}

// ------------------------------------------------------------------------------------------
// Stage / terminal costs and the mean-LCB objective of one candidate from the stored trajectory (one wavefront, lanes over the
// H + 1 time steps): the body of traj_cost_kernel (rollout.hip) and of the cost slice of the few-candidate gradient launch.
__device__ __forceinline__ void traj_cost_body(int c, int lane, const double* __restrict__ mu, const double* __restrict__ Sig,
                                                       const double* __restrict__ actions, const double* __restrict__ cost,
                                                       int D, int A, int H, double kappa, int clip, int use_constraints,
                                                       double* __restrict__ cm_out, double* __restrict__ cv_out,
                                                       double* __restrict__ J_out) {
    const int DA = D + A;
    const double* target = cost;
    const double* W = cost + DA;
    const double* WT = W + DA * DA;
    const double* smin = WT + D * D;
    const double* smax = smin + D;
    double jsum = 0.0;
    for (int t = lane; t <= H; t += 64) {
        const bool terminal = (t == H);
        const int n = terminal ? D : DA;
        const double* Wm = terminal ? WT : W;
        const double* m = mu + ((size_t)c * (H + 1) + t) * D;
        const double* S = Sig + ((size_t)c * (H + 1) + t) * D * D;
        const double* a = actions + ((size_t)c * H + (terminal ? 0 : t)) * A;
        auto err = [&](int i) { return (i < D ? m[i] : a[i - D]) - target[i]; };
        double cm = 0.0, cv = 0.0;
        // tr(Sigma W) and e^T W e
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) cm = fma(S[i * D + j], Wm[j * n + i], cm);
        for (int i = 0; i < n; ++i) {
            const double ei = err(i);
            for (int j = 0; j < n; ++j) cm = fma(ei * Wm[i * n + j], err(j), cm);
        }
        // tr(2 TS TS) with TS = W Sigma (state block), 4 (W^T e)^T Sigma (W e)
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) {
                double tij = 0.0, tji = 0.0;
                for (int k = 0; k < D; ++k) {
                    tij = fma(Wm[i * n + k], S[k * D + j], tij);
                    tji = fma(Wm[j * n + k], S[k * D + i], tji);
                }
                cv = fma(2.0 * tij, tji, cv);
                double v1 = 0.0, v2 = 0.0;
                for (int k = 0; k < n; ++k) {
                    v1 = fma(err(k), Wm[k * n + i], v1);
                    v2 = fma(Wm[j * n + k], err(k), v2);
                }
                cv = fma(4.0 * v1 * S[i * D + j], v2, cv);
            }
        if (use_constraints && !terminal) {
            for (int d = 0; d < D; ++d) {
                const double sg = S[d * D + d];              // the reference passes the VARIANCE as sigma (:63-64)
                cm += 0.5 * (1.0 + erf((smin[d] - m[d]) / (sg * 1.4142135623730951)))
                    + (1.0 - 0.5 * (1.0 + erf((smax[d] - m[d]) / (sg * 1.4142135623730951))));
            }
        }
        double ucb = -cm + kappa * sqrt(cv);
        if (clip) ucb = fmin(ucb, 0.0);
        jsum -= ucb;
        if (cm_out) cm_out[(size_t)c * (H + 1) + t] = cm;
        if (cv_out) cv_out[(size_t)c * (H + 1) + t] = cv;
    }
    // fixed-order sum over lanes (lane l holds steps l, l + 64, ...)
    for (int off = 32; off >= 1; off >>= 1) jsum += __shfl_xor(jsum, off, 64);
    if (lane == 0 && J_out) J_out[c] = jsum / (double)(H + 1);
}

}  // namespace gpmpc_hip
