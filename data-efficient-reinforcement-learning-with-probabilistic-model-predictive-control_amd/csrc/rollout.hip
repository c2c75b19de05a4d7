// rollout.hip -- host-side dispatch of the rollout kernel (tiling choice, LDS sizing) and the
// argmin kernel.  The kernel itself lives in rollout_kernel.h / rollout_dp.hip.
#include "rollout_stream_kernel.h"

namespace gpmpc_hip {

int launch_rollout_dp2(Handle*, RolloutArgs&, int, bool, size_t, hipStream_t);
int launch_rollout_dp3(Handle*, RolloutArgs&, int, bool, size_t, hipStream_t);
int launch_rollout_dp4(Handle*, RolloutArgs&, int, bool, size_t, hipStream_t);
int launch_rollout_dp6(Handle*, RolloutArgs&, int, bool, size_t, hipStream_t);
int launch_rollout_dp8(Handle*, RolloutArgs&, int, bool, size_t, hipStream_t);
int launch_rollout_dp16(Handle*, RolloutArgs&, int, bool, size_t, hipStream_t);

// ------------------------------------------------------------------------------------------
// Stage / terminal costs and the mean-LCB objective of every candidate, from the stored trajectory:
// setpoint_distance_reward_mapper.py:36-66 (stage), :135-141 (terminal); gp_mpc_controller.py:270-276.
// One wavefront per candidate, lanes over the H + 1 time steps; O(B H D^3) flops, negligible next to the
// rollout kernel, and keeping it out of the horizon loop is what lets that kernel fit 128 VGPRs.
__global__ __launch_bounds__(64) void traj_cost_kernel(const double* __restrict__ mu, const double* __restrict__ Sig,
                                                       const double* __restrict__ actions, const double* __restrict__ cost,
                                                       int D, int A, int H, double kappa, int clip, int use_constraints,
                                                       double* __restrict__ cm_out, double* __restrict__ cv_out,
                                                       double* __restrict__ J_out) {
    traj_cost_body(blockIdx.x, threadIdx.x, mu, Sig, actions, cost, D, A, H, kappa, clip, use_constraints, cm_out, cv_out, J_out);
}

// ------------------------------------------------------------------------------------------
// Keep-the-best rule of gp_mpc_controller.py:146-148 over a vector (single workgroup).
__global__ __launch_bounds__(1024) void argmin_kernel(const double* J, int B, long long first, double* out, int index_as_double,
                                                     const double* actions, int HA) {
    __shared__ long long s_best;
    __shared__ double s_v[16];
    __shared__ long long s_i[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double bv = INFINITY;
    long long bi = -1;
    for (int i = tid; i < B; i += 1024) {
        const double v = J[i];
        if (v < bv) { bv = v; bi = i; }          // strided scan keeps the lowest index per thread
    }
    for (int off = 32; off >= 1; off >>= 1) {
        const double ov = __shfl_xor(bv, off, 64);
        const long long oi = __shfl_xor(bi, off, 64);
        if (oi >= 0 && (bi < 0 || ov < bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
    }
    if (lane == 0) { s_v[wave] = bv; s_i[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w) {
            const double ov = s_v[w];
            const long long oi = s_i[w];
            if (oi >= 0 && (bi < 0 || ov < bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
        }
        if (first == 0 && B > 0 && J[0] != J[0]) { bv = J[0]; bi = 0; }   // NaN in global slot 0 is adopted and stays
        if (bi < 0) bv = INFINITY;                              // nothing selectable in this shard
        out[0] = bv;
        const long long gi = (bi < 0) ? -1 : bi + first;
        if (index_as_double) out[1] = (double)gi;
        else reinterpret_cast<long long*>(out)[1] = gi;
        s_best = bi < 0 ? 0 : bi;
    }
    if (actions) {                      // append the winner's action sequence to the record
        __syncthreads();
        const long long w = s_best;
        for (int k = tid; k < HA; k += 1024) out[2 + k] = actions[w * HA + k];
    }
}

// Zeroes an exchange buffer of the cooperative form with WRITE-THROUGH (agent-scope, sc1) stores: every later write to these
// buffers from another XCD is write-through as well, and no dirty zero line may stay behind in some XCD's L2 to be written back
// over it.
__global__ __launch_bounds__(256) void xch_zero_kernel(unsigned long long* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        __hip_atomic_store(p + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct InlineActs { double v[kInlineActs]; };
__global__ __launch_bounds__(64) void inline_acts_kernel(double* dst, int n, const InlineActs a) {
    for (int i = threadIdx.x; i < n; i += 64) dst[i] = a.v[i];
}

static int zero_exchange(Handle* h, unsigned long long* p, size_t words, hipStream_t s) {
    const unsigned blocks = (unsigned)((words + 255) / 256 < 256 ? (words + 255) / 256 : 256);
    hipLaunchKernelGGL(xch_zero_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, p, words);
    GPMPC_HIP_CHECK(h, hipGetLastError());
    return GPMPC_OK;
}

int launch_traj_cost(Handle* h, const RolloutArgs& a, double* cm, double* cv, double* J, hipStream_t s) {
    hipLaunchKernelGGL(traj_cost_kernel, dim3(a.B), dim3(64), 0, s, a.mu_out, a.Sig_out, a.actions, a.cost, a.D, a.A, a.H, a.kappa,
                       a.clip, a.use_constraints, cm, cv, J);
    GPMPC_HIP_CHECK(h, hipGetLastError());
    return GPMPC_OK;
}

int launch_argmin_to(Handle* h, const double* J, int B, long long first, const double* actions, int HA, double* out_dev,
                     hipStream_t s) {
    hipLaunchKernelGGL(argmin_kernel, dim3(1), dim3(1024), 0, s, J, B, first, out_dev, 1, actions, HA);
    GPMPC_HIP_CHECK(h, hipGetLastError());
    return GPMPC_OK;
}

int launch_argmin(Handle* h, const double* J, int B, long long first, hipStream_t s) {
    int rc = grow(h, h->best, 2);
    if (rc) return rc;
    hipLaunchKernelGGL(argmin_kernel, dim3(1), dim3(1024), 0, s, J, B, first, h->best.p, 0, (const double*)nullptr, 0);
    GPMPC_HIP_CHECK(h, hipGetLastError());
    return GPMPC_OK;
}

// Monomial exponent tables (graded order) for the separable evaluation, D <= 4.
static int ensure_monomials(Handle* h, int D) {
    if (h->mono_D == D) return GPMPC_OK;
    h->sep_kmax = 0; h->mono_CM = 0;
    for (int k = 0; k < 16; ++k) h->mono_cum[k] = 0;
    h->mono_D = D;
    if (D > 4) return GPMPC_OK;
    static thread_local int ex[kMaxMono * 4];
    static thread_local double wt[kMaxMono];
    int count = 0, kmax = -1;
    for (int k = 0; k <= kMaxTaylor; ++k) {
        // count monomials of exact degree k in D variables
        long long nk = 1;
        for (int i = 1; i < D; ++i) nk = nk * (k + i) / i;
        if (count + nk > kMaxMono) break;
        int a[4] = {0, 0, 0, 0};
        // enumerate compositions of k into D parts, lexicographic
        a[0] = k;
        for (;;) {
            double w = 1.0;
            for (int d = 0; d < 4; ++d) { ex[count * 4 + d] = a[d]; for (int e = 2; e <= a[d]; ++e) w /= (double)e; }
            wt[count++] = w;
            // next composition
            int i = D - 2;
            while (i >= 0 && a[i] == 0) --i;
            if (i < 0) break;
            a[i] -= 1;
            int tail = a[D - 1];
            a[D - 1] = 0;
            a[i + 1] = tail + 1;
            if (D == 1) break;
        }
        h->mono_cum[k] = count;
        kmax = k;
    }
    for (int k = kmax + 1; k < 16; ++k) h->mono_cum[k] = count;
    int rc = grow(h, h->mono_w, kMaxMono);
    if (rc) return rc;
    if (!h->mono_exp) GPMPC_HIP_CHECK(h, hipMalloc(&h->mono_exp, kMaxMono * 4 * sizeof(int)));
    GPMPC_HIP_CHECK(h, hipMemcpy(h->mono_exp, ex, count * 4 * sizeof(int), hipMemcpyHostToDevice));
    GPMPC_HIP_CHECK(h, hipMemcpy(h->mono_w.p, wt, count * sizeof(double), hipMemcpyHostToDevice));
    h->sep_kmax = kmax < 1 ? 0 : kmax;
    h->mono_CM = count;
    return GPMPC_OK;
}

static int padded_dim(int D) {
    const int sizes[] = {2, 3, 4, 6, 8, 16};
    for (int v : sizes) if (D <= v) return v;
    return -1;
}

int launch_rollout(Handle* h, RolloutArgs& a, hipStream_t s) {
    const int N = a.N, D = a.D, A = a.A, E = a.E;
    const int P = D * (D + 1) / 2;
    const int DP = padded_dim(D);
    if (DP < 0) { h->err = "D exceeds GPMPC_MAX_D"; return GPMPC_ERR_LIMIT; }

    // workgroup size: one candidate per workgroup.  Measured on MI355X (config 2..4 shapes, B = 256..2048):
    // 16 waves per candidate beat 8 even when the batch oversubscribes the 256 CUs (340k vs 265k
    // rollouts/s at B = 2048), so 1024 threads unless asked otherwise.
    int nt = h->opt_threads;
    if (nt != 256 && nt != 512 && nt != 1024) {
        nt = 1024;
        // Small memories with batches of several workgroups per CU: the pairwise pass of a step is a few hundred elements per
        // thread at most, the step is barrier- and latency-bound, and narrower workgroups overlap each other's serial phases
        // (N = 50, H = 15, tools/gpu_bench_ab.sh: B = 512: 1.85 / 2.61 / 2.16 M rollouts/s with 1024 / 512 / 256 threads,
        // B = 2048: 2.03 / 3.08 / 3.18 M, B = 8192: 2.08 / 3.27 / 3.51 M; at N = 200 1024 threads win at every B).
        if (N <= 64) {
            if (a.B >= 8 * h->num_cu) nt = 256;
            else if (a.B >= 2 * h->num_cu) nt = 512;
        }
    }
    // Mid-size memories with many workgroups per CU: two workgroups of 8 wavefronts share a CU (half the LDS each: the row
    // records of half the output pairs at a time), so that one's serial phases run beside the other's pairwise pass.  Measured
    // (round 4, profiles/r04h_share_cu.txt, config-2 shape N = 200): B = 512 equal, B = 1024 +2.6 %, B = 4096 +7.9 % (606 -> 654 k
    // rollouts/s); N = 500: -4 % at B = 1024, +2.6 % at 4096; N = 50: equal.  Hence from 8 workgroups per CU on, N in (64, 256].
    bool share_cu = false;
    if (h->opt_threads == 0 && h->opt_lds_kb == 0 && N > 64 && N <= 256 && D <= 4 && a.B >= 8 * h->num_cu && h->opt_pair_tiles != 1) {
        nt = 512;
        share_cu = true;
    }
    const int nw = nt / 64;
    int rcm = ensure_monomials(h, D);
    if (rcm) return rcm;
    const int CM = (h->opt_force_path == 0) ? h->mono_CM : 0;
    a.xrange = h->xrange.p; a.mono_exp = h->mono_exp; a.mono_w = h->mono_w.p;
    for (int k = 0; k < 16; ++k) a.mono_cum[k] = h->mono_cum[k];
    a.sep_kmax = (h->opt_force_path == 0) ? h->sep_kmax : 0;
    a.CM = CM;
    a.force_path = h->opt_force_path;
    a.force_sep = h->opt_force_sep;
    a.exact_dim = 0;

    // LDS budget of a workgroup of the fused-horizon kernel: all of it, or (option "lds_limit_kb", sharing rule below) a share
    // that lets two workgroups of 8 wavefronts live on one CU
    size_t lds_cap = (size_t)h->lds_limit;
    if (h->opt_lds_kb > 0 && (size_t)h->opt_lds_kb * 1024 < lds_cap) lds_cap = (size_t)h->opt_lds_kb * 1024;
    if (share_cu) lds_cap = (size_t)h->lds_limit / 2;
    // choose pairs-per-group G and row chunking for the LDS-resident variant
    bool gs = h->opt_force_global != 0;
    int G = 0, CH = 0, RC = 0;
    size_t lds_bytes = 0;
    // two adjacent columns per lane where the one-column loop is bound by the LDS broadcast bandwidth (small D)
    const bool cols2 = !gs && (h->opt_cols_per_lane == 2 || (h->opt_cols_per_lane == 0 && DP <= 4));
    // Batch-major pass for the diagonal pairs (pair_tile_kernel.h) when the D tables T_a no longer stay in an XCD's 4 MiB L2
    // and the batch is large enough to amortise a tile over many candidates; the fused-horizon kernel otherwise.
    bool tiled = false;
    if (!gs && cols2 && DP <= 4 && nt == 1024 && h->opt_pair_tiles != 2) {
        const double tri_bytes = 4.0 * D * (double)N * N;          // upper triangles of the D tables
        tiled = h->opt_pair_tiles == 1 || (tri_bytes >= 6.0e6 && a.B >= 2 * h->num_cu);
        tiled = tiled && tile_path_supported(h, a);
    }
    const int Pg = tiled ? (P - D > 0 ? P - D : 1) : P;           // pairs the per-candidate kernel keeps row records for
    const int NCu = cols2 ? (N + 1) / 2 : N;     // column units per row chunk
    bool cl_chunks = false;          // cooperative form: short row chunks (many items: its wavefronts are spread over several CUs)
    auto chunking = [&](int g) {
        // row chunks of up to 64 rows (fewer, longer items amortise the per-item prologue: 0.76 vs 0.80 ms at
        // config 2), but at least ~2 items per wave so that the queue can balance
        long long want = (cols2 ? 4LL : 2LL) * nw * 64;      // two-column items are twice as heavy: aim at twice as many
        long long rc = (want + (long long)g * NCu - 1) / ((long long)g * NCu);
        // longest chunk: 64 rows, or 48 in the two-column form -- once the queue walks a list of real items (round 5) shorter
        // chunks cost less than they did and fill the last round of the 16 wavefronts better (N = 500, D = 2: 36 items of 64 rows
        // = 2.25 rounds against 46 of 48 rows = 2.9; measured 206 -> 213 k rollouts/s, profiles/r05i_chunk_and_share_cu_sweeps.txt)
        const int chmax = cols2 ? 48 : 64;
        const long long rc64 = (N + chmax - 1) / chmax;
        if (rc < rc64) rc = rc64;
        int maxrc = (N + 15) / 16;
        if (rc > maxrc) rc = maxrc;
        if (rc < 1) rc = 1;
        // Cooperative form: an item of a lone wavefront is latency-bound (set-up ~2 k cycles, a trip of 2 rows ~400: 32 rows take
        // ~10 k cycles whether 1 or 4 wavefronts share the SIMD), so the members' 8 x cluster wavefronts want about one short item
        // each.  Measured at B = 1 (profiles/r06_cluster_sweep.txt): N = 200: 16 rows 0.216 ms against 0.222 (8) and 0.234 (32);
        // N = 500: 32 rows 0.495 against 0.557 (16) at 16 members.  The chunk length depends on nothing but N (not on B or the
        // cluster size): every few-candidate launch sums in the same order.
        if (cl_chunks) rc = (N + (N < 384 ? 16 : 32) - 1) / (N < 384 ? 16 : 32);
        if (h->opt_rows_per_chunk > 0) rc = (N + h->opt_rows_per_chunk - 1) / h->opt_rows_per_chunk;
        CH = (N + (int)rc - 1) / (int)rc;
        CH = (CH + 3) & ~3;                      // rows are processed in groups of 4
        if (CH > 64) CH = 64;                    // T_a carries 64 zero padding rows
        RC = (N + CH - 1) / CH;
    };
    // Few candidates (the reference's own regime: restarts_optim 1-2, one candidate per objective evaluation,
    // gp_mpc_controller.py:125-141): a cluster of workgroups per candidate (rollout_kernel<..., CL>) while clusters x candidates
    // fit the chip one workgroup per CU -- every member must be resident, they wait for each other once per horizon step.
    // Candidates are placed in groups of 8 (one per XCD), so the grid is 8 x cluster x ceil(B / 8) workgroups.
    int cluster = 1;
    const int groups8 = (a.B + 7) / 8;
    bool want_cluster = !gs && !tiled && cols2 && DP <= 4 && !share_cu && h->opt_lds_kb == 0 &&
                        h->opt_cluster != 1 && a.H < 8191 && 8 * groups8 * 2 <= h->num_cu;
    auto choose_layout = [&](bool cl) {
        G = 0;
        // X^T in LDS when it is small next to the budget (<= 32 KiB) and the layout still fits
        for (int xl = ((size_t)E * N * 8 <= 32 * 1024) ? 1 : 0; xl >= 0 && G == 0; --xl) {
            // (at most 48 pairs per group: one wavefront builds the step's work-item list, a lane per pair and per mean sum)
            for (int g = Pg < 48 ? Pg : 48; g >= 1; --g) {
                chunking(g);
                const int wpp = (RC * NCu + 63) / 64;
                Layout L = make_layout(N, D, A, E, g, DP, wpp, CM, CH, a.H * A, xl != 0, cl);
                if ((size_t)L.lds_total * 8 <= lds_cap) {
                    G = g; lds_bytes = (size_t)L.lds_total * 8; a.x_in_lds = xl;
                    break;
                }
            }
            // prefer all pairs in one group over the X^T copy
            if (G != 0 && G < (Pg < 48 ? Pg : 48) && xl == 1) { G = 0; }
        }
    };
    int cl_slots = 0;
    if (!gs && want_cluster) {
        // the cooperative form keeps all pairs in ONE group (one exchange per step) but a member holds per-point records only of the
        // pairs it owns items of (ClusterMap): its chunk length depends on N alone, so chunks, slots and cluster size come first
        cl_chunks = true;
        chunking(P);
        const int wpp_c = (RC * NCu + 63) / 64;
        int tri = 0;             // slots of a diagonal pair's triangle (the kernel's s_tri): the element-wise items that are always there
        for (int r = 0; r < RC; ++r) { const int first = (r * CH) / 2; tri += first < NCu ? NCu - first : 0; }
        const int wtri = (tri + 63) / 64;
        // ~4 element-wise items per member, and at least P_off + 2 D members (9 or more) where the chip has them for every candidate:
        // the separable pairs then sit on members of their own and nobody's per-point pass has more than two problems.  With that
        // many members the form pays from D^2 wtri ~ 18 on (one candidate, ms per rollout, cooperative / plain: D = 3: N = 50
        // 0.128 / 0.128, N = 80 0.129 / 0.151, N = 130 0.133 / 0.179; D = 2: N = 100 0.117 / 0.103, N = 150 equal, N = 200 0.139 / 0.173;
        // D = 1: N = 200 0.118 / 0.112, N = 400 0.139 / 0.203; D = 4: N = 100 0.184 / 0.334); with fewer members per candidate from
        // ~24 items on, as before
        const int cap = h->num_cu / (8 * groups8);
        const int items = D * wtri, P_off = D * (D - 1) / 2;
        const int cs_floor = (P_off + 2 * D > 9) ? P_off + 2 * D : 9;
        const bool roomy = cap >= cs_floor;
        int cs = 1;
        if (h->opt_cluster >= 2) cs = h->opt_cluster;
        else if (items >= 24 || (roomy && items * D >= 18)) {
            cs = (items + 3) / 4;
            if (roomy && cs < cs_floor) cs = cs_floor;
        }
        if (cs > 32) cs = 32;
        if (cs > cap) cs = cap;
        // (the pairs a member holds records of depend on how the cluster size divides the pairs: when the size of the rule does not
        //  fit the LDS, the next smaller ones are tried -- down to half of it; a size asked for by option is taken or not at all)
        for (const int cs_min = h->opt_cluster >= 2 ? cs : (cs + 1) / 2; cs >= 2 && cs >= cs_min && cluster == 1; --cs) {
            ClusterMap& cmap = a.cmap;
            cmap.plan(cs, D, wpp_c, wtri);
            cl_slots = 0;
            for (int m = 0; m < cs; ++m) { const int n = cmap.slots_needed(m); if (n > cl_slots) cl_slots = n; }
            for (int xl = ((size_t)E * N * 8 <= 32 * 1024) ? 1 : 0; xl >= 0 && cluster == 1; --xl) {
                const Layout L = make_layout(N, D, A, E, P, DP, wpp_c, CM, CH, a.H * A, xl != 0, true, cl_slots);
                if ((size_t)L.lds_total * 8 <= lds_cap) { G = P; lds_bytes = (size_t)L.lds_total * 8; a.x_in_lds = xl; cluster = cs; }
            }
        }
        // members of 8 wavefronts (no register spills: 145 VGPRs) from 8 members on; few, wide members otherwise (measured: B = 64,
        // 4 members: 0.267 / 0.270 ms with 1024 / 512 threads; B = 128, 2 members: 0.323 / 0.354; 16 members: 0.240 / 0.216)
        if (cluster > 1 && h->opt_threads == 0) nt = cluster >= 8 ? 512 : 1024;
        if (cluster == 1) cl_chunks = false;
    }
    if (!gs && cluster == 1) {
        choose_layout(false);
        if (G == 0) { gs = true; tiled = false; }
    }
    if (gs) {
        // large-N variant (rollout_stream_kernel.h): column factors + a double-buffered 64-row stage in LDS
        // matrix-core pair pass (DP = 8 / 16): chunks of 64 / 128 / 256 rows = 4 / 8 / 16 row tiles per (chunk, 16-column
        // block) item -- an item's set-up (column operands, addresses) is ~100 instructions and one L2 round trip, so long
        // items matter; rows past the data are zero records.  Option "rows_per_chunk" overrides (tests, A/B).
        if (DP % 4 == 0 && DP >= 8) {
            CH = (N >= 1024) ? 256 : 64;
            if (h->opt_rows_per_chunk > 0) CH = h->opt_rows_per_chunk <= 64 ? 64 : (h->opt_rows_per_chunk <= 128 ? 128 : 256);
        } else {
            CH = (N >= 64) ? 64 : ((N + 3) & ~3);
        }
        RC = (N + CH - 1) / CH;
        G = 1;
        StreamLayout SL = make_stream_layout(N, D, A, E, DP, CH, a.H * A);
        lds_bytes = (size_t)SL.total * 8;
        if (lds_bytes > (size_t)h->lds_limit) { h->err = "rollout: N too large for the LDS column-factor array"; return GPMPC_ERR_LIMIT; }
        nt = 1024;
    }
    a.G = G; a.CH = CH; a.RC = RC;
    a.cols2 = (cols2 && !gs) ? 1 : 0;
    if (a.act_inline_n > 0 && (gs || tiled)) {
        // only the fused-horizon kernel reads the sequence from its argument block: the other paths get it as its own launch
        InlineActs u;
        for (int i = 0; i < kInlineActs; ++i) u.v[i] = a.act_inline[i];
        hipLaunchKernelGGL(inline_acts_kernel, dim3(1), dim3(64), 0, s, a.act_store, a.act_inline_n, u);
        GPMPC_HIP_CHECK(h, hipGetLastError());
        a.act_inline_n = 0;
    }
    const int NCf = a.cols2 ? NCu : N;
    {
        // exact division by multiply-high: for d >= 2, umulhi(x, ceil(2^32 / d)) == x / d whenever x * d < 2^32
        // (magic 0 encodes d == 1: no division)
        const unsigned wppv = (unsigned)((RC * NCf + 63) / 64);
        auto magic = [](unsigned d) -> unsigned { return d <= 1 ? 0u : (unsigned)((0x100000000ULL + d - 1) / d); };
        a.magic_N = magic((unsigned)NCf);
        a.magic_wpp = magic(wppv);
        a.magic_pt = magic((unsigned)N);
        if ((unsigned long long)RC * NCf * NCf >= 0x100000000ULL || (unsigned long long)G * wppv * wppv >= 0x100000000ULL ||
            (!gs && (unsigned long long)(D + 2 * P) * N * N >= 0x100000000ULL)) {      // magic_pt: the fused-horizon kernel only
            h->err = "rollout: index range too large for the multiply-high division"; return GPMPC_ERR_LIMIT;
        }
        if (!gs && (unsigned long long)G * wppv >= 65000ULL) {     // 16-bit entries of the step's work-item list
            h->err = "rollout: too many work-item slots for the 16-bit item list"; return GPMPC_ERR_LIMIT;
        }
    }

    // the propagation always stores the trajectory (the cost kernel reads it back): own buffers if
    // the caller did not ask for them
    double* user_cm = a.cm_out;
    double* user_cv = a.cv_out;
    double* user_J = a.J_out;
    {
        const size_t nmu = (size_t)a.B * (a.H + 1) * D, nS = nmu * D;
        if (!a.mu_out || !a.Sig_out) {
            int rc = grow(h, h->traj, nmu + nS);
            if (rc) return rc;
            if (!a.mu_out) a.mu_out = h->traj.p;
            if (!a.Sig_out) a.Sig_out = h->traj.p + nmu;
        }
    }
    int rc = GPMPC_OK;
    auto launch_kernel = [&]() {
        switch (DP) {
            case 2:  return launch_rollout_dp2(h, a, nt, gs, lds_bytes, s);
            case 3:  return launch_rollout_dp3(h, a, nt, gs, lds_bytes, s);
            case 4:  return launch_rollout_dp4(h, a, nt, gs, lds_bytes, s);
            case 6:  return launch_rollout_dp6(h, a, nt, gs, lds_bytes, s);
            case 8:  return launch_rollout_dp8(h, a, nt, gs, lds_bytes, s);
            default: return launch_rollout_dp16(h, a, nt, gs, lds_bytes, s);
        }
    };
    a.cl_dbg = h->opt_cl_dbg;
    a.cl_slots = cl_slots;
    a.xch_uc = nullptr;
    a.cluster = cluster; a.xch = nullptr; a.xch_n = 0; a.xch_tag0 = 0;
    h->last_cluster = cluster;
    if (cluster > 1) {
        const unsigned wppv = (unsigned)((RC * NCu + 63) / 64);
        a.xch_n = G * (int)wppv + G + D * (D + 1) + 32;       // item slots | separable pairs | mean sums | the members' XCD ids
        const size_t words = (size_t)8 * groups8 * 4 * a.xch_n;
        // tags never repeat while the HANDLE lives (8192 per launch), short of the 32-bit wrap; the buffer is zeroed when it is
        // (re)allocated and at the wrap
        // The epoch survives a re-allocation: the new buffer may sit where the old one did, and lines of the old one -- with the
        // old launches' tags -- can still be in some XCD's L2 (a re-allocated buffer whose tags restarted at 1 let a member accept
        // such a line: the members' states then differ, so do their item lists, and somebody waits for a value nobody publishes:
        // seen as a bounded-wait timeout in round 6).  Only the 32-bit wrap restarts it, 2^19 launches later.
        const bool fresh = !h->xch.p || words > h->xch.cap;
        int rcx = grow(h, h->xch, words);
        if (rcx) return rcx;
        const bool wrap = h->xch_epoch >= (1u << 19) - 1;
        if (fresh || wrap) { rcx = zero_exchange(h, reinterpret_cast<unsigned long long*>(h->xch.p), h->xch.cap, s); if (rcx) return rcx; }
        if (words > h->xch_uc_cap) {
            if (h->xch_uc) GPMPC_HIP_CHECK(h, hipFree(h->xch_uc));
            h->xch_uc = nullptr; h->xch_uc_cap = 0;
            void* q = nullptr;
            if (hipExtMallocWithFlags(&q, words * sizeof(unsigned long long), hipDeviceMallocUncached) != hipSuccess) {
                (void)hipGetLastError();
                GPMPC_HIP_CHECK(h, hipExtMallocWithFlags(&q, words * sizeof(unsigned long long), hipDeviceMallocFinegrained));
            }
            h->xch_uc = reinterpret_cast<unsigned long long*>(q);
            h->xch_uc_cap = words;
            rcx = zero_exchange(h, h->xch_uc, words, s);
            if (rcx) return rcx;
        } else if (wrap) {
            rcx = zero_exchange(h, h->xch_uc, h->xch_uc_cap, s);
            if (rcx) return rcx;
        }
        if (wrap) h->xch_epoch = 1;
        a.xch_tag0 = h->xch_epoch * 8192u;
        h->xch_epoch += 1;
        a.xch = reinterpret_cast<unsigned long long*>(h->xch.p);
        a.xch_uc = h->xch_uc;
    }
    h->last_rollout_path = gs ? 1 : (tiled ? 2 : 0);
    // the batch-major state of a (possibly re-used) argument block is set on EVERY call, never inherited from an earlier one
    a.tiled = tiled ? 1 : 0;
    if (!tiled) a.grad_mom = nullptr;                      // only the batch-major forward can form the gradient's tile moments
    h->last_fused_tiles = (tiled && a.grad_mom) ? 1 : 0;
    if (!tiled) { a.t_begin = 0; a.t_end = 0; a.slow = nullptr; a.tile_part = nullptr; a.ntiles = 0; }
    if (tiled) {
        // per horizon step: parameters + batch-major tiles of the diagonal pairs, then the per-candidate rest of the step
        rc = tile_workspace(h, a);
        if (!rc) rc = launch_tile_state_init(h, a, s);
        for (int t = 0; t < a.H && !rc; ++t) {
            rc = launch_pair_tiles(h, a, t, s);
            a.t_begin = t; a.t_end = t + 1;
            if (!rc) rc = launch_kernel();
        }
    } else {
        rc = launch_kernel();
    }
    if (rc) return rc;
    if ((user_cm || user_cv || user_J) && !a.defer_cost) {
        hipLaunchKernelGGL(traj_cost_kernel, dim3(a.B), dim3(64), 0, s, a.mu_out, a.Sig_out, a.actions, a.cost, D, A, a.H,
                           a.kappa, a.clip, a.use_constraints, user_cm, user_cv, user_J);
        GPMPC_HIP_CHECK(h, hipGetLastError());
    }
    return GPMPC_OK;
}

}  // namespace gpmpc_hip
