// point_pass_kernel.h -- the O(N) part of one horizon step per candidate, for the batch-major path (gfx950 / MI355X).
//
// With the N x N work of the diagonal output pairs moved to pair_tile_kernel.h, what is left of a moment-matched step
// (reference rl_gp_mpc/control_objects/models/gp_model.py:112-180) is linear in the number of memory points:
//   * the mean part: lb_a = exp(-q_a / 2) beta_a, M_a, V_a                                   (:140-153)
//   * the off-diagonal output pairs in their separable form (rollout_kernel.h):
//       sum_ij ra_i rb_j exp(g_i . w_j) = sum_|alpha|<=K (sum_i ra_i g_i^alpha) (sum_j rb_j w_j^alpha) / alpha!
//   * the D x D update of the state distribution                                            (:105-108, 176-178)
// The fused-horizon kernel does this with 1024 threads per candidate, phase by phase behind workgroup barriers, its
// monomial sums through scalar exponent loops (one branch per factor): at config 4 that part alone took 1.57 ms per
// horizon step of 2048 candidates (rocprofv3, round 3) -- as much as the whole N x N pass.  Here:
//   * step_params_kernel does ALL the small algebra of a step up front, one thread per problem (D mean problems,
//     D (D + 1) / 2 pair problems per candidate): no serial phase is left in the per-candidate kernel;
//   * point_pass_kernel: one workgroup of 4 wavefronts per candidate, two per CU; the wavefronts pull tasks
//     (mean part of output a | (pair, side, band of monomials)) from an LDS counter; a task is ONE pass over the
//     points with lanes owning points, the per-point factors computed on the fly (nothing per-point is stored) and
//     up to 36 monomial sums in registers, the monomials of a band enumerated at COMPILE time (one FMA each);
//   * bands: the monomials of degree <= K in D variables are split by a host-side recursion into products
//     x^e * {monomials of degree <= m in the variables s .. D-1} of at most 36 members (SepTable);
//   * candidates that have an off-diagonal pair outside the separable range at this step (direct exp, degree beyond
//     the table) are left to the element-wise kernel (rollout_kernel<.., TILED>), which runs after this one and
//     skips everybody else: `slow[c] == t + 1` is the hand-over.
// Fixed summation order everywhere (lane partials -> wave_reduce16 / 8 -> ordered sums): bitwise reproducible, batch-independent.
#pragma once
#include "rollout_kernel.h"

namespace gpmpc_hip {

constexpr int kSepCap = 36;          // monomial sums a task keeps in registers (72 VGPRs; with 72 sums the D = 4 kernel spilled 145)
constexpr int kSepAcc = (kSepCap + 7) & ~7;      // accumulator array of a task: kSepCap rounded up to the reduction groups
constexpr int kSepMaxBands = 64;

struct SepBand {
    int nv, m, s, cnt, off;          // tail: monomials of degree <= m in the nv variables s .. s + nv - 1; cnt members at `off`
    int e[4];                        // prefix exponents
};

struct SepTable {
    int ks;                          // highest degree with a separable form
    int nb[16], first[16], total[16], woff[16];      // per degree: bands, first band, monomials, offset of its weights
    SepBand band[kSepMaxBands];
};

struct StepArgs {
    const double* Xt;       // (E, N)
    const double* beta;     // (D, N)
    const double* Tm;       // (D, N + kTPad, N)
    const double* ils2;     // (D, E)
    const double* var;      // (D)
    const double* logvar;   // (D)
    const double* xrange;   // (2, E)
    const double* actions;  // (B, H, A)
    double* mu;             // (B, H + 1, D)     trajectory: the step reads index t and writes t + 1
    double* Sig;            // (B, H + 1, D, D)
    double* crec;           // (B, CS) step record of a candidate: input mean | mean problems | pair problems
    double* part;           // (B, D, ntiles) partial sums of the diagonal pairs (pair_tile_kernel)
    double* pout;           // (B, PO) what the point pass hands to step_combine_kernel: off-diagonal pair totals (P slots) | mean sums D (D + 1)
    int* slow;              // (B) t + 1 when the candidate's step t goes to the element-wise kernel
    const SepTable* septab;
    const double* sepw;     // 1 / alpha! in band order, per degree
    int N, D, A, E, H, B, t;
    int include_time;
    double time0;
    int nb, ntiles;         // tile rows / columns, upper-triangle tile count nb (nb + 1) / 2
    int cch, nchunk;        // candidates per tile workgroup, chunks
    int CS, off_mean, off_pair, PR, PRP; // record layout: PR = DP * DP + 2 doubles per mean problem, PRP = 4 DP * DP + 2 per pair
    int force_path;         // 1: direct exp for every pair, 2: never separable (tests)
    int ksep;               // highest separable degree (septab->ks)
    int mom_stride;         // doubles per (pair, side) moment array in LDS
    int PO;                 // doubles per candidate in pout
    int item0;              // TRAJ records: first (candidate, step) item of this launch
    int compact;            // 1: records hold the D diagonal pair problems only (TRAJ); 0: the forward step's full records
    int fused_t;            // >= 0: the gradient's tile moments are formed inside the forward's step `fused_t` (items = candidates)
};

__host__ __device__ inline int pair_index(int a, int b, int D) { return a * D - (a * (a - 1)) / 2 + (b - a); }

// ------------------------------------------------------------------------------------------
// All the D x D algebra of step t, one thread per problem: phase P1 of rollout_kernel.
//   mean problem a : A_a = Sigma + diag(l_a^2) -> A_a^-1, c_a = var_a / sqrt(det B_a)                 (gp_model.py:141-150)
//   pair problem q : R = Sigma diag(1/l_a^2 + 1/l_b^2) + I -> Z = R^-1 Sigma, 1 / sqrt(det R), Taylor degree   (:156-163, 176)
// TRAJ: the records of ALL (candidate, step) items of a stored trajectory for the gradient's tile pass (pair_tile_grad_kernel.h):
// p.B counts items, item = candidate * H + step, only the diagonal pair problems are formed.
template <int DP, bool TRAJ = false>
__global__ __launch_bounds__(256) void step_params_kernel(const StepArgs p) {
    const int D = p.D, E = p.E, A = p.A;
    const int P = D * (D + 1) / 2;
    const int per = D + P;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)p.B * per) return;
    const int c = (int)(idx / per), j = (int)(idx - (long long)c * per);
    const int cc = TRAJ ? (p.item0 + c) / p.H : c;          // candidate and step of the trajectory arrays
    const int tt = TRAJ ? (p.item0 + c) - cc * p.H : p.t;
    const double* mu = p.mu + ((size_t)cc * (p.H + 1) + tt) * D;
    const double* Sg = p.Sig + ((size_t)cc * (p.H + 1) + tt) * D * D;
    double* rec = p.crec + (size_t)c * p.CS;
    if (j == 0) {
        for (int e = 0; e < E; ++e) {
            double v;
            if (e < D) v = mu[e];
            else if (e < D + A) v = p.actions[((size_t)cc * p.H + tt) * A + (e - D)];
            else v = p.time0 + (double)tt;
            rec[e] = v;
        }
    }
    if (TRAJ && j < D) return;
    double m[DP][2 * DP];
    if (j < D) {
        const int a = j;
        const double* il = p.ils2 + (size_t)a * E;
        double prodil = 1.0;
#pragma unroll
        for (int i = 0; i < DP; ++i) {
            const double il2 = (i < D) ? il[i] : 1.0;
            prodil *= il2;
#pragma unroll
            for (int k = 0; k < DP; ++k) {
                const bool in = (i < D && k < D);
                m[i][k] = (in ? Sg[i * D + k] : 0.0) + (i == k ? 1.0 / il2 : 0.0);
                m[i][DP + k] = (in && i == k) ? 1.0 : 0.0;
            }
        }
        const double detA = small_solve<DP>(m);
        double* out = rec + p.off_mean + a * p.PR;
#pragma unroll
        for (int i = 0; i < DP; ++i)
#pragma unroll
            for (int k = 0; k < DP; ++k) out[i * DP + k] = (i < D && k < D) ? m[i][DP + k] : 0.0;
        out[DP * DP] = p.var[a] / sqrt(detA * prodil);
        out[DP * DP + 1] = 0.0;
        return;
    }
    const int q = j - D;
    int a = 0, rem = q;
    while (rem >= D - a) { rem -= D - a; ++a; }
    const int b = a + rem;
    if (TRAJ && a != b) return;
    const double* ila = p.ils2 + (size_t)a * E;
    const double* ilb = p.ils2 + (size_t)b * E;
#pragma unroll
    for (int i = 0; i < DP; ++i)
#pragma unroll
        for (int k = 0; k < DP; ++k) {
            const bool in = (i < D && k < D);
            const double sg = in ? Sg[i * D + k] : 0.0;
            const double dab = in ? ila[k] + ilb[k] : 0.0;
            m[i][k] = sg * dab + (i == k ? 1.0 : 0.0);
            m[i][DP + k] = sg;
        }
    double ur[DP], wr[DP];
#pragma unroll
    for (int i = 0; i < DP; ++i) {
        const double mi = (i < D) ? mu[i] : 0.0;
        const double rg = (i < D) ? fmax(fabs(p.xrange[i] - mi), fabs(p.xrange[E + i] - mi)) : 0.0;
        ur[i] = (i < D) ? rg * ila[i] : 0.0;
        wr[i] = (i < D) ? rg * ilb[i] : 0.0;
    }
    const double detR = small_solve<DP>(m);
    double* out = rec + p.off_pair + (TRAJ ? a : q) * p.PRP;      // TRAJ records hold the D diagonal pairs only
    double cmax = 0.0;
#pragma unroll
    for (int i = 0; i < DP; ++i) {
        double r = 0.0;
#pragma unroll
        for (int k = 0; k < DP; ++k) {
            const double z = (i < D && k < D) ? m[i][DP + k] : 0.0;
            out[i * DP + k] = z;
            r = fma(fabs(z), wr[k], r);
        }
        cmax += r * ur[i];
    }
    int K = 0;
    if (p.force_path != 1 && cmax <= kTaylorMaxArg[kMaxTaylor]) {
        K = 1;
#pragma unroll
        for (int k = 1; k < kMaxTaylor; ++k) K += (cmax > kTaylorMaxArg[k]) ? 1 : 0;
    }
    if (a != b) {
        if (K >= 1 && K <= p.ksep && p.force_path == 0) K |= 64;          // separable: point_pass_kernel
        else atomicMax(p.slow + c, p.t + 1);                             // element-wise: rollout_kernel<.., TILED>
    }
    out[DP * DP] = 1.0 / sqrt(detR);
    out[DP * DP + 1] = (double)K;
    // What the per-point passes need of this pair, so that a point costs two small matrix-vector products:
    //   log-factor of a point  k' = log var_c - (nu^T Q_c nu + action / time terms) / 2,  Q_c = L_c - L_c sym(Z) L_c,  L_c = diag(1 / l_c^2)
    //   (= k + u^T Z u / 2 of rollout_kernel.h with u = L_c nu; c = a on the row side, b on the column side), and  g = Z^T L_a nu = G nu.
    double* Q0 = out + DP * DP + 2;
    double* G = Q0 + DP * DP;
    double* Q1 = G + DP * DP;
#pragma unroll
    for (int i = 0; i < DP; ++i)
#pragma unroll
        for (int k = 0; k < DP; ++k) {
            const bool in = (i < D && k < D);
            const double zs = in ? 0.5 * (m[i][DP + k] + m[k][DP + i]) : 0.0;
            const double lai = in ? ila[i] : 0.0, lak = in ? ila[k] : 0.0, lbi = in ? ilb[i] : 0.0, lbk = in ? ilb[k] : 0.0;
            Q0[i * DP + k] = (i == k ? lai : 0.0) - lai * zs * lak;
            Q1[i * DP + k] = (i == k ? lbi : 0.0) - lbi * zs * lbk;
            G[i * DP + k] = in ? m[k][DP + i] * lak : 0.0;               // g_i = sum_k Z_ki l_ak^-2 nu_k
        }
}

// The trajectory's index 0 and the hand-over flags (the step kernels read their state from the trajectory arrays).
__global__ __launch_bounds__(256) void step_state_init_kernel(const RolloutArgs p, int* slow) {
    const int D = p.D;
    const size_t per = (size_t)D + (size_t)D * D;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx < (size_t)p.B) slow[idx] = 0;
    if (idx >= (size_t)p.B * per) return;
    const size_t c = idx / per;
    const int k = (int)(idx - c * per);
    if (k < D) p.mu_out[c * (p.H + 1) * D + k] = p.mu0[k];
    else p.Sig_out[c * (p.H + 1) * D * D + (k - D)] = p.S0[k - D];
}

// ------------------------------------------------------------------------------------------
// acc[n] += base * (n-th monomial of degree <= M in the NV tail variables), nested order: exponent of the first variable
// outermost.  Everything unrolls: acc indices are compile-time, one FMA per monomial (powers of the last variable tabulated).
template <int M>
__device__ inline void tail1(double (&acc)[kSepAcc], double base, double t0) {
    double pwr = base;
#pragma unroll
    for (int l = 0; l <= M; ++l) { acc[l] += pwr; pwr *= t0; }
}

template <int M>
__device__ inline void tail2(double (&acc)[kSepAcc], double base, double t0, double t1) {
    double pw[M + 1];
    pw[0] = 1.0;
#pragma unroll
    for (int l = 1; l <= M; ++l) pw[l] = pw[l - 1] * t1;
    int n = 0;
    double p0 = base;
#pragma unroll
    for (int i = 0; i <= M; ++i) {
#pragma unroll
        for (int l = 0; l <= M - i; ++l) { acc[n] = fma(p0, pw[l], acc[n]); ++n; }
        p0 *= t0;
    }
}

template <int M>
__device__ inline void tail3(double (&acc)[kSepAcc], double base, double t0, double t1, double t2) {
    double pw[M + 1];
    pw[0] = 1.0;
#pragma unroll
    for (int l = 1; l <= M; ++l) pw[l] = pw[l - 1] * t2;
    int n = 0;
    double p0 = base;
#pragma unroll
    for (int i = 0; i <= M; ++i) {
        double p1 = p0;
#pragma unroll
        for (int j = 0; j <= M - i; ++j) {
#pragma unroll
            for (int l = 0; l <= M - i - j; ++l) { acc[n] = fma(p1, pw[l], acc[n]); ++n; }
            p1 *= t1;
        }
        p0 *= t0;
    }
}

template <int M>
__device__ inline void tail4(double (&acc)[kSepAcc], double base, double t0, double t1, double t2, double t3) {
    double pw[M + 1];
    pw[0] = 1.0;
#pragma unroll
    for (int l = 1; l <= M; ++l) pw[l] = pw[l - 1] * t3;
    int n = 0;
    double p0 = base;
#pragma unroll
    for (int i = 0; i <= M; ++i) {
        double p1 = p0;
#pragma unroll
        for (int j = 0; j <= M - i; ++j) {
            double p2 = p1;
#pragma unroll
            for (int k = 0; k <= M - i - j; ++k) {
#pragma unroll
                for (int l = 0; l <= M - i - j - k; ++l) { acc[n] = fma(p2, pw[l], acc[n]); ++n; }
                p2 *= t2;
            }
            p1 *= t1;
        }
        p0 *= t0;
    }
}

// code = nv * 16 + m (wave-uniform).  The host only emits bands this switch knows (sep_band_supported).
template <int DP>
__device__ inline void tail_dispatch(int code, double (&acc)[kSepAcc], double base, double t0, double t1, double t2, double t3) {
    // every acc index below is a compile-time constant: a run-time index would move the sums to scratch memory
    if ((code & 15) == 0) { acc[0] += base; return; }
    switch (code) {
        case 1 * 16 + 1: tail1<1>(acc, base, t0); break;
        case 1 * 16 + 2: tail1<2>(acc, base, t0); break;
        case 1 * 16 + 3: tail1<3>(acc, base, t0); break;
        case 1 * 16 + 4: tail1<4>(acc, base, t0); break;
        case 1 * 16 + 5: tail1<5>(acc, base, t0); break;
        case 1 * 16 + 6: tail1<6>(acc, base, t0); break;
        case 1 * 16 + 7: tail1<7>(acc, base, t0); break;
        case 1 * 16 + 8: tail1<8>(acc, base, t0); break;
        case 1 * 16 + 9: tail1<9>(acc, base, t0); break;
        case 1 * 16 + 10: tail1<10>(acc, base, t0); break;
        case 1 * 16 + 11: tail1<11>(acc, base, t0); break;
        case 1 * 16 + 12: tail1<12>(acc, base, t0); break;
        case 1 * 16 + 13: tail1<13>(acc, base, t0); break;
        case 1 * 16 + 14: tail1<14>(acc, base, t0); break;
        case 2 * 16 + 1: tail2<1>(acc, base, t0, t1); break;
        case 2 * 16 + 2: tail2<2>(acc, base, t0, t1); break;
        case 2 * 16 + 3: tail2<3>(acc, base, t0, t1); break;
        case 2 * 16 + 4: tail2<4>(acc, base, t0, t1); break;
        case 2 * 16 + 5: tail2<5>(acc, base, t0, t1); break;
        case 2 * 16 + 6: tail2<6>(acc, base, t0, t1); break;
        case 2 * 16 + 7: tail2<7>(acc, base, t0, t1); break;
        default: break;
    }
    if constexpr (DP >= 3) {
        switch (code) {
            case 3 * 16 + 1: tail3<1>(acc, base, t0, t1, t2); break;
            case 3 * 16 + 2: tail3<2>(acc, base, t0, t1, t2); break;
            case 3 * 16 + 3: tail3<3>(acc, base, t0, t1, t2); break;
            case 3 * 16 + 4: tail3<4>(acc, base, t0, t1, t2); break;
            default: break;
        }
    }
    if constexpr (DP >= 4) {
        switch (code) {
            case 4 * 16 + 1: tail4<1>(acc, base, t0, t1, t2, t3); break;
            case 4 * 16 + 2: tail4<2>(acc, base, t0, t1, t2, t3); break;
            case 4 * 16 + 3: tail4<3>(acc, base, t0, t1, t2, t3); break;
            default: break;
        }
    }
}

__host__ __device__ inline long long sep_binom(int n, int k) {
    long long r = 1;
    for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;
    return r;
}

// bands the device code can evaluate (tail_dispatch)
__host__ __device__ inline bool sep_band_supported(int nv, int m) {
    if (m < 0 || m > kMaxTaylor) return false;
    if (nv <= 1 || m == 0) return true;
    if (nv == 2) return m <= 7;
    if (nv == 3) return m <= 4;
    if (nv == 4) return m <= 3;
    return false;
}

struct PointLayout {
    int rec, tab, exptab, ils2, logvar, var, mu, Sig, s1, M, Vs, Sp, mom, task, act, total;     // offsets in doubles
};

__host__ __device__ inline PointLayout make_point_layout(int N, int D, int E, int CS, int mom_stride) {
    const int P = D * (D + 1) / 2, Poff = P - D;
    PointLayout L;
    int o = 0;
    L.exptab = o; o += 64;
    L.rec = o;    o += rnd2(CS);
    L.tab = o;    o += rnd2((int)((sizeof(SepTable) + 7) / 8));
    L.ils2 = o;   o += rnd2(D * E);
    L.logvar = o; o += rnd2(D);
    L.var = o;    o += rnd2(D);
    L.mu = o;     o += rnd2(D);
    L.Sig = o;    o += rnd2(D * D);
    L.s1 = o;     o += rnd2(D * (D + 1));
    L.M = o;      o += rnd2(D);
    L.Vs = o;     o += rnd2(D * D);
    L.Sp = o;     o += rnd2(P);
    L.task = o;   o += 80;                    // counter, count, up to 150 tasks (ints)
    L.mom = o;    o += (Poff > 0 ? Poff : 1) * 2 * mom_stride;
    L.act = o;    o += D * rnd2(N);           // action / time terms of the log-factors, per output and point
    L.total = o;
    return L;
}

// ------------------------------------------------------------------------------------------
template <int DP>
__global__ __launch_bounds__(256, 2) void point_pass_kernel(const StepArgs p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int NT = 256, NW = 4;
    const int c = blockIdx.x;
    if (p.slow[c] == p.t + 1) return;                 // this candidate's step is the element-wise kernel's
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = p.N, D = p.D, E = p.E;
    const PointLayout L = make_point_layout(N, D, E, p.CS, p.mom_stride);
    double* s_exptab = smem + L.exptab;
    double* s_rec = smem + L.rec;
    const SepTable* s_tab = reinterpret_cast<const SepTable*>(smem + L.tab);
    double* c_ils2 = smem + L.ils2;
    double* c_logvar = smem + L.logvar;
    double* s_s1 = smem + L.s1;
    double* s_mom = smem + L.mom;
    double* s_act = smem + L.act;
    const int NA = rnd2(N);
    int* s_task = reinterpret_cast<int*>(smem + L.task);      // [0] counter, [1] count, [2..] tasks

    // ---- stage the candidate's step record and the small tables ------------------------------------------------------
    {
        const double* rec = p.crec + (size_t)c * p.CS;
        for (int i = tid; i < p.CS; i += NT) s_rec[i] = rec[i];
        const int* tsrc = reinterpret_cast<const int*>(p.septab);
        int* tdst = reinterpret_cast<int*>(smem + L.tab);
        for (int i = tid; i < (int)(sizeof(SepTable) / 4); i += NT) tdst[i] = tsrc[i];
        for (int i = tid; i < 64; i += NT) s_exptab[i] = kExp2Tab[i];
        for (int i = tid; i < D * E; i += NT) c_ils2[i] = p.ils2[i];
        for (int i = tid; i < D; i += NT) {
            c_logvar[i] = p.logvar[i];
        }
    }
    __syncthreads();
    // action / time part of every log-factor: sum_{e >= D} (x_e - m_e)^2 / l_ce^2, once per (output, point) instead of once per task
    for (int i = tid; i < D * N; i += NT) {
        const int co = i / N, pt = i - co * N;
        double v = 0.0;
        for (int e = D; e < E; ++e) {
            const double dx = p.Xt[(size_t)e * N + pt] - s_rec[e];
            v = fma(dx * dx, c_ils2[co * E + e], v);
        }
        s_act[co * NA + pt] = v;
    }
    if (tid == 0) {
        // tasks: [kind 0: mean part of output a] [kind 1: (pair q, side, band)]; the long ones (pairs) first
        int n = 0;
        int q = 0;
        for (int a = 0; a < D; ++a)
            for (int b = a; b < D; ++b, ++q) {
                if (a == b) continue;
                const int K = (int)s_rec[p.off_pair + q * p.PRP + DP * DP + 1] & 63;
                for (int side = 0; side < 2; ++side)
                    for (int bi = 0; bi < s_tab->nb[K]; ++bi) s_task[2 + n++] = (1 << 24) | (q << 16) | (side << 8) | (s_tab->first[K] + bi);
            }
        for (int a = 0; a < D; ++a) s_task[2 + n++] = a;
        s_task[0] = 0;
        s_task[1] = n;
    }
    __syncthreads();
    const int ntask = s_task[1];
    const double* mo = s_rec;                                     // input mean of the step

    auto pull = [&]() -> int {
        int v = 0;
        if (lane == 0) v = __hip_atomic_fetch_add(&s_task[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return __builtin_amdgcn_readfirstlane(v);
    };
    const int niter = (N + 63) >> 6;
    for (int ti = pull(); ti < ntask; ti = pull()) {
        const int task = __builtin_amdgcn_readfirstlane(s_task[2 + ti]);
        if ((task >> 24) == 0) {
            // ---- mean part of output a: s1[a] = sum_p lb_p [1, nu_p]  (gp_model.py:140-153) -----------------------------
            const int a = task;
            const double* Ai = s_rec + p.off_mean + a * p.PR;
            double Am[DP][DP];
#pragma unroll
            for (int i = 0; i < DP; ++i)
#pragma unroll
                for (int k = 0; k < DP; ++k) Am[i][k] = Ai[i * DP + k];
            double a0 = 0.0, ad[DP];
#pragma unroll
            for (int d = 0; d < DP; ++d) ad[d] = 0.0;
            for (int it = 0; it < niter; ++it) {
                const int pt0 = it * 64 + lane;
                const bool live = pt0 < N;
                const int pt = live ? pt0 : N - 1;
                double nu[DP];
#pragma unroll
                for (int d = 0; d < DP; ++d) nu[d] = (d < D) ? p.Xt[(size_t)d * N + pt] - mo[d] : 0.0;
                double qv = 0.0;
#pragma unroll
                for (int i = 0; i < DP; ++i) {
                    double r = 0.0;
#pragma unroll
                    for (int k = 0; k < DP; ++k) r = fma(Am[i][k], nu[k], r);
                    qv = fma(nu[i], r, qv);
                }
                qv += s_act[a * NA + pt];
                double lb = fast_exp(-0.5 * qv, s_exptab) * p.beta[(size_t)a * N + pt];
                lb = live ? lb : 0.0;
                a0 += lb;
#pragma unroll
                for (int d = 0; d < DP; ++d) ad[d] = fma(lb, nu[d], ad[d]);
            }
            const double t0 = wave_sum(a0);
            if (lane == 0) s_s1[a * (D + 1)] = t0;
#pragma unroll
            for (int d = 0; d < DP; ++d) {
                const double td = wave_sum(ad[d]);
                if (lane == 0 && d < D) s_s1[a * (D + 1) + 1 + d] = td;
            }
            continue;
        }
        // ---- (pair, side, band): moments  sum_p wt_p x_p^alpha  over the band's monomials ---------------------------------
        const int q = (task >> 16) & 255, side = (task >> 8) & 255, bi = task & 255;
        int a = 0, rem = q;
        while (rem >= D - a) { rem -= D - a; ++a; }
        const int b = a + rem;
        const int co = side ? b : a;                           // the output whose lengthscales scale nu
        const double* pr = s_rec + p.off_pair + q * p.PRP;
        const double* Qs = pr + DP * DP + 2 + (side ? 2 * DP * DP : 0);       // Q of this side
        const double* Gs = pr + 2 * DP * DP + 2;                               // g = G nu (row side)
        double Qm[DP][DP], Xm[DP][DP];                                        // Xm: the map nu -> monomial variables (G | diag(1 / l_b^2))
#pragma unroll
        for (int i = 0; i < DP; ++i)
#pragma unroll
            for (int k = 0; k < DP; ++k) {
                Qm[i][k] = Qs[i * DP + k];
                Xm[i][k] = side ? ((i == k && i < D) ? c_ils2[co * E + i] : 0.0) : Gs[i * DP + k];
            }
        const double lv = c_logvar[co];
        const double* actc = s_act + co * NA;
        const SepBand* bd = &s_tab->band[bi];
        const int bs = __builtin_amdgcn_readfirstlane(bd->s);
        const int code = __builtin_amdgcn_readfirstlane(bd->nv * 16 + bd->m);
        const int cnt = __builtin_amdgcn_readfirstlane(bd->cnt);
        const int boff = __builtin_amdgcn_readfirstlane(bd->off);
        int pe[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) pe[d] = __builtin_amdgcn_readfirstlane(bd->e[d]);
        double acc[kSepAcc];
#pragma unroll
        for (int n = 0; n < kSepAcc; ++n) acc[n] = 0.0;
        // the state coordinates and beta of the next 64 points travel while the current ones are worked on
        double xc[DP], bc;
        {
            const int pt = lane < N ? lane : N - 1;
#pragma unroll
            for (int d = 0; d < DP; ++d) xc[d] = (d < D) ? p.Xt[(size_t)d * N + pt] : 0.0;
            bc = p.beta[(size_t)co * N + pt];
        }
        for (int it = 0; it < niter; ++it) {
            const int pt0 = it * 64 + lane;
            const bool live = pt0 < N;
            const int pt = live ? pt0 : N - 1;
            double xn[DP], bn;
            {
                const int ptn = (pt0 + 64 < N) ? pt0 + 64 : N - 1;
#pragma unroll
                for (int d = 0; d < DP; ++d) xn[d] = (d < D) ? p.Xt[(size_t)d * N + ptn] : 0.0;
                bn = p.beta[(size_t)co * N + ptn];
            }
            double nu[DP], xq[DP];
#pragma unroll
            for (int d = 0; d < DP; ++d) nu[d] = (d < D) ? xc[d] - mo[d] : 0.0;
            double qf = actc[pt];
#pragma unroll
            for (int i = 0; i < DP; ++i) {
                double r = 0.0, xi = 0.0;
#pragma unroll
                for (int k = 0; k < DP; ++k) {
                    r = fma(Qm[i][k], nu[k], r);
                    xi = fma(Xm[i][k], nu[k], xi);
                }
                qf = fma(nu[i], r, qf);
                xq[i] = xi;
            }
            const double kkv = fma(-0.5, qf, lv);
            double wt = fast_exp(kkv, s_exptab) * bc;
            wt = live ? wt : 0.0;
            double x[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) x[d] = (d < DP) ? xq[d < DP ? d : 0] : 0.0;
            // prefix monomial (wave-uniform exponents)
#pragma unroll
            for (int d = 0; d < 4; ++d)
                for (int e = 0; e < pe[d]; ++e) wt *= x[d];
            const double t0 = bs == 0 ? x[0] : (bs == 1 ? x[1] : (bs == 2 ? x[2] : x[3]));
            const double t1 = bs == 0 ? x[1] : (bs == 1 ? x[2] : x[3]);
            const double t2 = bs == 0 ? x[2] : x[3];
            const double t3 = x[3];
            tail_dispatch<DP>(code, acc, wt, t0, t1, t2, t3);
#pragma unroll
            for (int d = 0; d < DP; ++d) xc[d] = xn[d];
            bc = bn;
        }
        double* mom = s_mom + (size_t)(((q - a - 1) * 2) + side) * p.mom_stride + boff;       // (q - a - 1): index among the off-diagonal pairs
        // lane partials over the wavefront: 16 + 16 + 8 values (wave_reduce*: permlane swaps and DPP, no LDS round trips);
        // lane 4 m ends with the total of value m of its group
        static_assert(kSepCap <= 40, "reduction groups below");
        {
            const double(&g0)[16] = *reinterpret_cast<const double(*)[16]>(&acc[0]);
            const double t0r = wave_reduce16(g0);
            if ((lane & 3) == 0 && (lane >> 2) < cnt) mom[lane >> 2] = t0r;
            if (cnt > 16) {
                const double(&g1)[16] = *reinterpret_cast<const double(*)[16]>(&acc[16]);
                const double t1r = wave_reduce16(g1);
                if ((lane & 3) == 0 && 16 + (lane >> 2) < cnt) mom[16 + (lane >> 2)] = t1r;
            }
            if (cnt > 32) {
                const double(&g2)[8] = *reinterpret_cast<const double(*)[8]>(&acc[32]);
                const double t2r = wave_reduce8(g2);
                if ((lane & 3) == 0 && lane < 32 && 32 + (lane >> 2) < cnt) mom[32 + (lane >> 2)] = t2r;
            }
        }
    }
    __syncthreads();

    // ---- totals of the off-diagonal pairs (x 1 / sqrt(det R)) and the mean sums -> HBM for step_combine_kernel ---------
    {
        double* po = p.pout + (size_t)c * p.PO;
        int q = 0;
        for (int a = 0; a < D; ++a)
            for (int b = a; b < D; ++b, ++q) {
                if (a == b || (q & (NW - 1)) != wave) continue;
                const double* pr = s_rec + p.off_pair + q * p.PRP;
                const int K = (int)pr[DP * DP + 1] & 63;
                const int C = s_tab->total[K];
                const double* wgt = p.sepw + s_tab->woff[K];
                const double* Gm = s_mom + (size_t)((q - a - 1) * 2) * p.mom_stride;
                const double* Wm = Gm + p.mom_stride;
                double v = 0.0;
                for (int n = lane; n < C; n += 64) v = fma(Gm[n] * Wm[n], wgt[n], v);
                v = wave_sum(v);
                if (lane == 0) po[q] = v * pr[DP * DP];
            }
        const int P = D * (D + 1) / 2;
        for (int i = tid; i < D * (D + 1); i += NT) po[P + i] = s_s1[i];
    }
}

// ------------------------------------------------------------------------------------------
// The D x D end of a step, one wavefront per candidate, once the tiles (diagonal pairs) and the point pass (mean sums,
// off-diagonal pairs) of the step are there: M, V, S, Sigma_{t+1} = S + Sigma + C + C^T, mu_{t+1} = mu + M
// (gp_model.py:105-108, 152-153, 176-178).  Candidates of the element-wise kernel are skipped.
template <int DP>
__global__ __launch_bounds__(64) void step_combine_kernel(const StepArgs p) {
    __shared__ double s_Sp[DP * (DP + 1) / 2], s_M[DP], s_Vs[DP * DP], s_Sig[DP * DP];
    const int c = blockIdx.x, lane = threadIdx.x;
    if (p.slow[c] == p.t + 1) return;
    const int D = p.D;
    const int P = D * (D + 1) / 2;
    const double* rec = p.crec + (size_t)c * p.CS;
    const double* po = p.pout + (size_t)c * p.PO;
    const double* s1 = po + P;
    for (int i = lane; i < D * D; i += 64) s_Sig[i] = p.Sig[((size_t)c * (p.H + 1) + p.t) * D * D + i];
    {
        int q = 0;
        for (int a = 0; a < D; ++a)
            for (int b = a; b < D; ++b, ++q) {
                if (a == b) {
                    // sum of the tiles' partial sums in a fixed order (i <= j only: factor 2)
                    const double* tp = p.part + ((size_t)c * D + a) * p.ntiles;
                    double v = 0.0;
                    for (int k = lane; k < p.ntiles; k += 64) v += tp[k];
                    v = wave_sum(v);
                    if (lane == 0) s_Sp[q] = 2.0 * v * rec[p.off_pair + q * p.PRP + DP * DP];
                } else if (lane == 0) {
                    s_Sp[q] = po[q];
                }
            }
    }
    if (lane < D) s_M[lane] = rec[p.off_mean + lane * p.PR + DP * DP] * s1[lane * (D + 1)];                           // M_a (:152)
    for (int idx = lane; idx < D * D; idx += 64) {
        const int k = idx / D, a = idx - k * D;
        const double* Ai = rec + p.off_mean + a * p.PR;
        double s = 0.0;
        for (int j = 0; j < D; ++j) s = fma(Ai[k * DP + j], s1[a * (D + 1) + 1 + j], s);
        s_Vs[idx] = Ai[DP * DP] * s;                                                                              // state rows of V (:153)
    }
    wave_lds_sync();
    for (int idx = lane; idx < D * D; idx += 64) {
        const int i = idx / D, j = idx - i * D;
        const int a = i < j ? i : j, b = i < j ? j : i;
        const double S = s_Sp[pair_index(a, b, D)] - s_M[i] * s_M[j] + (i == j ? p.var[i] : 0.0);
        double cij = 0.0, cji = 0.0;
        for (int k = 0; k < D; ++k) {
            cij = fma(s_Sig[i * D + k], s_Vs[k * D + j], cij);
            cji = fma(s_Sig[j * D + k], s_Vs[k * D + i], cji);
        }
        p.Sig[((size_t)c * (p.H + 1) + (p.t + 1)) * D * D + idx] = S + s_Sig[idx] + (cij + cji);
    }
    if (lane < D) p.mu[((size_t)c * (p.H + 1) + (p.t + 1)) * D + lane] = p.mu[((size_t)c * (p.H + 1) + p.t) * D + lane] + s_M[lane];
}

}  // namespace gpmpc_hip
