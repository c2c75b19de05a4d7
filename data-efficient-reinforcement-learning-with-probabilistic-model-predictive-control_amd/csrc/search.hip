// search.hip -- device-resident candidate optimisation.
//
// The reference optimises the action sequence with `restarts_optim` sequential scipy L-BFGS-B solves, every
// evaluation one Python call of compute_mean_lcb_trajectory (gp_mpc_controller.py:125-141).  gpmpc_cem_search is a
// cross-entropy search over the same box [0, 1]^(H A) whose whole loop stays on the GPU: per iteration
//     cem_sample_kernel   B optimiser vectors around the current mean / std (iteration 0: uniform; slot 0 = the incumbent,
//                         or the caller's warm start), counter-based Philox draws (or draws supplied by the caller),
//                         and the action mapper (identity, or scaled deltas + cumulative sum + clamp:
//                         actions_mappers/derivative_action_mapper.py:28-35) -> model actions (B, H, A)
//     rollout launch      the hot path: objective of the B sequences
//     cem_refit_kernel    one workgroup: NaN -> +inf, bitonic sort of (J, index) (ties by index = numpy's stable
//                         argsort), incumbent update, mean / population std (+ 1e-3) of the elites per coordinate
// are enqueued back to back on the caller's stream; nothing is read back between iterations.  The caller synchronises
// once, for the best vector and its objective.
#include "gpmpc_internal.h"

namespace gpmpc_hip {

namespace {

constexpr int kCemMaxB = 4096;

__device__ inline void philox_round(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3, unsigned k0, unsigned k1) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

// Philox4x32-10: two uniform doubles in (0, 1) from (counter, key)
__device__ inline void philox_uniform2(unsigned long long seed, unsigned a, unsigned b, unsigned c, double& u0, double& u1) {
    unsigned c0 = a, c1 = b, c2 = c, c3 = 0x9E3779B9u;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    const unsigned long long x0 = ((unsigned long long)c0 << 21) ^ (unsigned long long)(c1 >> 11);      // 53 bits
    const unsigned long long x1 = ((unsigned long long)c2 << 21) ^ (unsigned long long)(c3 >> 11);
    u0 = ((double)(x0 & ((1ull << 53) - 1)) + 0.5) * (1.0 / 9007199254740992.0);
    u1 = ((double)(x1 & ((1ull << 53) - 1)) + 0.5) * (1.0 / 9007199254740992.0);
}

// one thread per (candidate, coordinate): optimiser vectors of this iteration
// (`b0`, `B_total`: this launch draws candidates [b0, b0 + B) of a population of B_total -- the counters and the caller's
// noise are indexed by the GLOBAL candidate, so the union of the slices of several GPUs is the single-GPU population)
__global__ __launch_bounds__(256) void cem_sample_kernel(int it, int B, int n, unsigned long long seed, const double* __restrict__ noise,
                                                         const double* __restrict__ mean, const double* __restrict__ stdv,
                                                         const double* __restrict__ best, int have_first,
                                                         const double* __restrict__ first, double* __restrict__ X, int b0, int B_total) {
    const int lidx = blockIdx.x * 256 + threadIdx.x;
    if (lidx >= B * n) return;
    const int b = lidx / n + b0, k = lidx - (lidx / n) * n;
    const int idx = b * n + k;
    double v;
    if (b == 0 && it > 0) v = best[k];                             // keep the incumbent
    else if (b == 0 && have_first) v = first[k];                   // warm start (previous solution shifted by one step)
    else {
        double draw;
        if (noise) draw = noise[(size_t)it * B_total * n + idx];
        else {
            double u0, u1;
            philox_uniform2(seed, (unsigned)idx, (unsigned)it, 0x243F6A88u, u0, u1);
            draw = (it == 0) ? u0 : sqrt(-2.0 * log(u0)) * cos(6.283185307179586 * u1);
        }
        v = (it == 0) ? draw : fmin(fmax(fma(stdv[k], draw, mean[k]), 0.0), 1.0);
    }
    X[lidx] = v;
}

// one thread per (candidate, action dimension): optimiser vector -> model actions
__global__ __launch_bounds__(256) void cem_map_kernel(int B, int H, int A, int mapper, const double* __restrict__ max_change,
                                                      const double* __restrict__ a_prev, const double* __restrict__ X,
                                                      double* __restrict__ actions) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * A) return;
    const int b = idx / A, a = idx - b * A;
    const double* x = X + (size_t)b * H * A + a;
    double* out = actions + (size_t)b * H * A + a;
    if (mapper == 0) {
        for (int t = 0; t < H; ++t) out[(size_t)t * A] = x[(size_t)t * A];
    } else {
        const double m = max_change[a];
        double s = a_prev[a];
        for (int t = 0; t < H; ++t) {
            s += x[(size_t)t * A] * 2.0 * m - m;
            out[(size_t)t * A] = fmin(fmax(s, 0.0), 1.0);        // pass-through clamp: the sum itself is not clamped
        }
    }
}

// single workgroup: sort, incumbent, elite statistics
__global__ __launch_bounds__(1024) void cem_refit_kernel(int B, int n, int n_elite, const double* __restrict__ J,
                                                         const double* __restrict__ X, double* __restrict__ mean,
                                                         double* __restrict__ stdv, double* __restrict__ best /* n + 1: x | J */,
                                                         int first_iteration) {
    __shared__ double key[kCemMaxB];
    __shared__ int val[kCemMaxB];
    const int tid = threadIdx.x;
    int P2 = 1;
    while (P2 < B) P2 <<= 1;
    for (int i = tid; i < P2; i += 1024) {
        double v = (i < B) ? J[i] : INFINITY;
        if (v != v) v = INFINITY;
        key[i] = v;
        val[i] = (i < B) ? i : 0x7fffffff;
    }
    __syncthreads();
    for (int size = 2; size <= P2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < P2 / 2; i += 1024) {
                const int lo = 2 * i - (i & (stride - 1));              // index with bit `stride` clear
                const int hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const double klo = key[lo], khi = key[hi];
                const int vlo = val[lo], vhi = val[hi];
                const bool gt = (klo > khi) || (klo == khi && vlo > vhi);
                if (gt == up) { key[lo] = khi; key[hi] = klo; val[lo] = vhi; val[hi] = vlo; }
            }
            __syncthreads();
        }
    }
    const double bestJ_old = first_iteration ? INFINITY : best[n];
    // the first iteration always adopts its leader (even when every objective is non-finite): `best` is then never read
    // uninitialised by the sampler of the next iteration
    const bool improve = first_iteration || key[0] < bestJ_old;
    const int ib = val[0];
    __syncthreads();                                       // every thread has read best[n] before thread 0 replaces it
    for (int k = tid; k < n; k += 1024) {
        double s = 0.0;
        for (int e = 0; e < n_elite; ++e) s += X[(size_t)val[e] * n + k];
        const double m = s / (double)n_elite;
        double q = 0.0;
        for (int e = 0; e < n_elite; ++e) { const double d = X[(size_t)val[e] * n + k] - m; q = fma(d, d, q); }
        mean[k] = m;
        stdv[k] = sqrt(q / (double)n_elite) + 1e-3;
        if (improve) best[k] = X[(size_t)ib * n + k];
    }
    if (tid == 0) best[n] = improve ? key[0] : bestJ_old;
}

// in-LDS bitonic sort of (key, val), ascending, ties by val; P2 = power of two >= the number of entries
__device__ inline void cem_sort(double* key, int* val, int P2, int tid) {
    for (int size = 2; size <= P2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < P2 / 2; i += 1024) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const double klo = key[lo], khi = key[hi];
                const int vlo = val[lo], vhi = val[hi];
                const bool gt = (klo > khi) || (klo == khi && vlo > vhi);
                if (gt == up) { key[lo] = khi; key[hi] = klo; val[lo] = vhi; val[hi] = vlo; }
            }
            __syncthreads();
        }
    }
}

// Candidates sharded over GPUs: the n_elite best of THIS slice as records [J (NaN -> +inf) | global index | optimiser vector (n)],
// sorted like numpy's stable argsort; a slice shorter than n_elite pads with (+inf, INT_MAX, 0 ...) records, which sort behind
// every real candidate.  One workgroup.
__global__ __launch_bounds__(1024) void cem_elites_kernel(int B, int n, int n_elite, int b0, const double* __restrict__ J,
                                                          const double* __restrict__ X, double* __restrict__ rec) {
    __shared__ double key[kCemMaxB];
    __shared__ int val[kCemMaxB];
    const int tid = threadIdx.x;
    int P2 = 1;
    while (P2 < B) P2 <<= 1;
    for (int i = tid; i < P2; i += 1024) {
        double v = (i < B) ? J[i] : INFINITY;
        if (v != v) v = INFINITY;
        key[i] = v;
        val[i] = (i < B) ? i : 0x7fffffff;
    }
    __syncthreads();
    cem_sort(key, val, P2, tid);
    const int RS = n + 2;
    for (int i = tid; i < n_elite * RS; i += 1024) {
        const int e = i / RS, c = i - e * RS;
        const bool real = e < B;
        double v;
        if (c == 0) v = real ? key[e] : INFINITY;
        else if (c == 1) v = real ? (double)(val[e] + b0) : 2147483647.0;
        else v = real ? X[(size_t)val[e] * n + (c - 2)] : 0.0;
        rec[i] = v;
    }
}

// ... and the refit of cem_refit_kernel on the union of the slices' elite records (R = lists x n_elite of them, gathered over
// RCCL): the same sort, incumbent rule and elite statistics, in the same summation order -- every GPU computes the same
// state, bit for bit the one the single-GPU search reaches on the same draws.  state = mean (n) | std (n) | best (n) | best J.
__global__ __launch_bounds__(1024) void cem_merge_kernel(int R, int n, int n_elite, const double* __restrict__ rec,
                                                         double* __restrict__ state, int first_iteration) {
    __shared__ double key[kCemMaxB];
    __shared__ int val[kCemMaxB];          // record slot
    __shared__ int gid[kCemMaxB];          // global candidate index of the slot (tie-break)
    const int tid = threadIdx.x;
    const int RS = n + 2;
    int P2 = 1;
    while (P2 < R) P2 <<= 1;
    // sort by (J, global index): the sort's tie-break is on `val`, so sort global indices and recover slots afterwards
    for (int i = tid; i < P2; i += 1024) {
        key[i] = (i < R) ? rec[(size_t)i * RS] : INFINITY;
        val[i] = (i < R) ? (int)rec[(size_t)i * RS + 1] : 0x7fffffff;
    }
    __syncthreads();
    cem_sort(key, val, P2, tid);
    // slot of each of the n_elite leaders (global indices are unique among real records)
    for (int e = tid; e < n_elite; e += 1024) {
        int slot = 0;
        for (int r = 0; r < R; ++r) if ((int)rec[(size_t)r * RS + 1] == val[e]) { slot = r; break; }
        gid[e] = slot;
    }
    __syncthreads();
    double* mean = state;
    double* stdv = state + n;
    double* best = state + 2 * n;
    const double bestJ_old = first_iteration ? INFINITY : best[n];
    const bool improve = first_iteration || key[0] < bestJ_old;
    const int ib = gid[0];
    __syncthreads();
    for (int k = tid; k < n; k += 1024) {
        double s = 0.0;
        for (int e = 0; e < n_elite; ++e) s += rec[(size_t)gid[e] * RS + 2 + k];
        const double m = s / (double)n_elite;
        double q = 0.0;
        for (int e = 0; e < n_elite; ++e) { const double d = rec[(size_t)gid[e] * RS + 2 + k] - m; q = fma(d, d, q); }
        mean[k] = m;
        stdv[k] = sqrt(q / (double)n_elite) + 1e-3;
        if (improve) best[k] = rec[(size_t)ib * RS + 2 + k];
    }
    if (tid == 0) best[n] = improve ? key[0] : bestJ_old;
}

}  // namespace

int run_cem_search(Handle* h, RolloutArgs& a, int iterations, int n_elite, unsigned long long seed, const double* first_host,
                   int mapper, const double* max_change_host, const double* a_prev_host, const double* noise_dev,
                   double* best_out_dev, hipStream_t s) {
    const int B = a.B, H = a.H, A = a.A, n = H * A;
    if (B < 2 || B > kCemMaxB || iterations < 1 || n_elite < 1 || n_elite > B) { h->err = "cem: need 2 <= B <= 4096, 1 <= n_elite <= B"; return GPMPC_ERR_ARG; }
    // workspace: X (B n) | actions (B n) | J (B) | mean (n) | std (n) | first (n) | mapper params (2 A)
    const size_t need = 2 * (size_t)B * n + B + 3 * (size_t)n + 2 * (size_t)A;
    int rc = grow(h, h->cemws, need);
    if (rc) return rc;
    double* X = h->cemws.p;
    double* acts = X + (size_t)B * n;
    double* J = acts + (size_t)B * n;
    double* mean = J + B;
    double* stdv = mean + n;
    double* first = stdv + n;
    double* mc = first + n;
    double* ap = mc + A;
    if (first_host) GPMPC_HIP_CHECK(h, hipMemcpyAsync(first, first_host, n * sizeof(double), hipMemcpyHostToDevice, s));
    if (mapper != 0) {
        if (!max_change_host || !a_prev_host) { h->err = "cem: the derivative mapper needs max_change and the previous action"; return GPMPC_ERR_ARG; }
        GPMPC_HIP_CHECK(h, hipMemcpyAsync(mc, max_change_host, A * sizeof(double), hipMemcpyHostToDevice, s));
        GPMPC_HIP_CHECK(h, hipMemcpyAsync(ap, a_prev_host, A * sizeof(double), hipMemcpyHostToDevice, s));
    }
    a.actions = acts;
    a.J_out = J;
    a.mu_out = nullptr; a.Sig_out = nullptr; a.cm_out = nullptr; a.cv_out = nullptr;
    for (int it = 0; it < iterations; ++it) {
        hipLaunchKernelGGL(cem_sample_kernel, dim3((B * n + 255) / 256), dim3(256), 0, s, it, B, n, seed, noise_dev, mean, stdv,
                           best_out_dev, first_host ? 1 : 0, first, X, 0, B);
        hipLaunchKernelGGL(cem_map_kernel, dim3((B * A + 255) / 256), dim3(256), 0, s, B, H, A, mapper, mc, ap, X, acts);
        GPMPC_HIP_CHECK(h, hipGetLastError());
        RolloutArgs ai = a;
        rc = launch_rollout(h, ai, s);
        if (rc) return rc;
        hipLaunchKernelGGL(cem_refit_kernel, dim3(1), dim3(1024), 0, s, B, n, n_elite, J, X, mean, stdv, best_out_dev, it == 0 ? 1 : 0);
        GPMPC_HIP_CHECK(h, hipGetLastError());
    }
    return GPMPC_OK;
}

// One iteration of the search for the slice [b0, b0 + Bl) of a population of B_total candidates (a.B = Bl): draws + mapper +
// rollout + the slice's elite records.  `state` = mean | std | best | best J as left by run_cem_merge of the previous iteration.
int run_cem_local(Handle* h, RolloutArgs& a, int B_total, int b0, int it, int n_elite, unsigned long long seed,
                  const double* first_host, int mapper, const double* max_change_host, const double* a_prev_host,
                  const double* noise_dev, const double* state_dev, double* elites_out_dev, hipStream_t s) {
    const int Bl = a.B, H = a.H, A = a.A, n = H * A;
    if (B_total < 2 || B_total > kCemMaxB || Bl < 0 || b0 < 0 || b0 + Bl > B_total || it < 0 || n_elite < 1 || n_elite > B_total) {
        h->err = "cem: need 2 <= B_total <= 4096, a slice inside it, 1 <= n_elite <= B_total"; return GPMPC_ERR_ARG;
    }
    const int RS = n + 2;
    if (Bl == 0) {                       // more GPUs than candidates: padding records only
        hipLaunchKernelGGL(cem_elites_kernel, dim3(1), dim3(1024), 0, s, 0, n, n_elite, b0, (const double*)nullptr, (const double*)nullptr,
                           elites_out_dev);
        GPMPC_HIP_CHECK(h, hipGetLastError());
        (void)RS;
        return GPMPC_OK;
    }
    const size_t need = 2 * (size_t)Bl * n + Bl + (size_t)n + 2 * (size_t)A;
    int rc = grow(h, h->cemws, need);
    if (rc) return rc;
    double* X = h->cemws.p;
    double* acts = X + (size_t)Bl * n;
    double* J = acts + (size_t)Bl * n;
    double* first = J + Bl;
    double* mc = first + n;
    double* ap = mc + A;
    const bool use_first = first_host && it == 0 && b0 == 0;
    if (use_first) GPMPC_HIP_CHECK(h, hipMemcpyAsync(first, first_host, n * sizeof(double), hipMemcpyHostToDevice, s));
    if (mapper != 0) {
        if (!max_change_host || !a_prev_host) { h->err = "cem: the derivative mapper needs max_change and the previous action"; return GPMPC_ERR_ARG; }
        GPMPC_HIP_CHECK(h, hipMemcpyAsync(mc, max_change_host, A * sizeof(double), hipMemcpyHostToDevice, s));
        GPMPC_HIP_CHECK(h, hipMemcpyAsync(ap, a_prev_host, A * sizeof(double), hipMemcpyHostToDevice, s));
    }
    a.actions = acts;
    a.J_out = J;
    a.mu_out = nullptr; a.Sig_out = nullptr; a.cm_out = nullptr; a.cv_out = nullptr;
    hipLaunchKernelGGL(cem_sample_kernel, dim3((Bl * n + 255) / 256), dim3(256), 0, s, it, Bl, n, seed, noise_dev, state_dev, state_dev + n,
                       state_dev + 2 * n, use_first ? 1 : 0, first, X, b0, B_total);
    hipLaunchKernelGGL(cem_map_kernel, dim3((Bl * A + 255) / 256), dim3(256), 0, s, Bl, H, A, mapper, mc, ap, X, acts);
    GPMPC_HIP_CHECK(h, hipGetLastError());
    RolloutArgs ai = a;
    rc = launch_rollout(h, ai, s);
    if (rc) return rc;
    hipLaunchKernelGGL(cem_elites_kernel, dim3(1), dim3(1024), 0, s, Bl, n, n_elite, b0, J, X, elites_out_dev);
    GPMPC_HIP_CHECK(h, hipGetLastError());
    return GPMPC_OK;
}

int run_cem_merge(Handle* h, const double* elites_dev, int lists, int n_elite, int n, int it, double* state_dev, hipStream_t s) {
    if (lists < 1 || n_elite < 1 || n < 1 || (long long)lists * n_elite > kCemMaxB) {
        h->err = "cem: lists x n_elite must not exceed 4096 records"; return GPMPC_ERR_LIMIT;
    }
    hipLaunchKernelGGL(cem_merge_kernel, dim3(1), dim3(1024), 0, s, lists * n_elite, n, n_elite, elites_dev, state_dev, it == 0 ? 1 : 0);
    GPMPC_HIP_CHECK(h, hipGetLastError());
    return GPMPC_OK;
}

}  // namespace gpmpc_hip
