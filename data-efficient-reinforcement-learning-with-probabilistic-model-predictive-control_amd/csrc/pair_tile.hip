// pair_tile.hip -- host side of the batch-major path: per horizon step  step_params_kernel (all D x D algebra) ->
// pair_tile_kernel (N x N work of the diagonal pairs, tiles x candidates) -> point_pass_kernel (O(N) work per candidate:
// mean part, separable off-diagonal pairs, state update); candidates outside the separable range go to the element-wise
// kernel launched after these by launch_rollout.  Workspace, tiling, the table of monomial bands, launches.
#include <vector>

#include "pair_tile_kernel.h"
#include "pair_tile_grad_kernel.h"

namespace gpmpc_hip {

static int tile_dp(int D) { return D <= 2 ? 2 : (D == 3 ? 3 : 4); }

// ------------------------------------------------------------------------------------------
// Bands of the separable evaluation (point_pass_kernel.h): the monomials of degree <= m in the variables s .. D-1, times the
// prefix x^e, are one band when there are at most kSepCap of them; otherwise split by the exponent of x_s (0 | >= 1).
static void decompose(int D, int s, int m, int (&e)[4], std::vector<SepBand>& out) {
    const int nv = D - s;
    const long long cnt = sep_binom(nv + m, nv);
    if (cnt <= kSepCap || nv == 1) {
        SepBand b{};
        b.nv = nv; b.m = m; b.s = s; b.cnt = (int)cnt; b.off = 0;
        for (int d = 0; d < 4; ++d) b.e[d] = e[d];
        out.push_back(b);
        return;
    }
    decompose(D, s + 1, m, e, out);
    e[s] += 1;
    decompose(D, s, m - 1, e, out);
    e[s] -= 1;
}

// weights 1 / alpha! of a band's monomials in the device's enumeration order (tail2 / tail3 / tail4: first tail variable outermost)
static void band_weights(const SepBand& b, std::vector<double>& w) {
    auto fact = [](int n) { double f = 1.0; for (int i = 2; i <= n; ++i) f *= i; return f; };
    int ex[4];
    std::vector<int> t(b.nv, 0);
    // nested loops over the tail exponents with total degree <= m, lexicographic with the first variable outermost
    std::vector<int> cur(b.nv, 0);
    for (;;) {
        for (int d = 0; d < 4; ++d) ex[d] = b.e[d];
        for (int k = 0; k < b.nv; ++k) ex[b.s + k] += cur[k];
        double v = 1.0;
        for (int d = 0; d < 4; ++d) v /= fact(ex[d]);
        w.push_back(v);
        // increment: innermost (last) variable first, bounded by the remaining degree
        int k = b.nv - 1;
        for (;;) {
            int used = 0;
            for (int i = 0; i < k; ++i) used += cur[i];
            if (cur[k] < b.m - used) { ++cur[k]; break; }
            cur[k] = 0;
            if (--k < 0) return;
        }
    }
}

static int ensure_sep_table(Handle* h, int D) {
    if (h->septab_D == D && h->septab) return GPMPC_OK;
    SepTable T{};
    std::vector<double> w;
    // highest degree whose monomial count stays within the LDS moment arrays (as the fused-horizon kernel: <= kMaxMono)
    int ks = 0, nb_total = 0;
    for (int K = 1; K <= kMaxTaylor && D >= 2; ++K) {
        if (sep_binom(D + K, D) > kMaxMono) break;
        std::vector<SepBand> bands;
        int e[4] = {0, 0, 0, 0};
        decompose(D, 0, K, e, bands);
        bool ok = nb_total + (int)bands.size() <= kSepMaxBands;
        for (const SepBand& b : bands) ok = ok && sep_band_supported(b.nv, b.m);
        if (!ok) break;
        T.first[K] = nb_total; T.nb[K] = (int)bands.size(); T.woff[K] = (int)w.size();
        int off = 0;
        for (SepBand& b : bands) {
            b.off = off;
            off += b.cnt;
            band_weights(b, w);
            T.band[nb_total++] = b;
        }
        T.total[K] = off;
        ks = K;
    }
    T.ks = ks;
    if (!h->septab) GPMPC_HIP_CHECK(h, hipMalloc(&h->septab, sizeof(SepTable)));
    GPMPC_HIP_CHECK(h, hipMemcpy(h->septab, &T, sizeof(SepTable), hipMemcpyHostToDevice));
    int rc = grow(h, h->sepw, w.size() + 1);
    if (rc) return rc;
    if (!w.empty()) GPMPC_HIP_CHECK(h, hipMemcpy(h->sepw.p, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice));
    h->septab_D = D;
    h->sep_ks = ks;
    h->sep_cmax = ks ? T.total[ks] : 0;
    return GPMPC_OK;
}

static void step_geometry(const Handle* h, const RolloutArgs& a, StepArgs& t) {
    const int DP = tile_dp(a.D);
    const int P = a.D * (a.D + 1) / 2;
    t.nb = (a.N + kTileW - 1) / kTileW;
    t.ntiles = t.nb * (t.nb + 1) / 2;
    t.PR = DP * DP + 2;
    t.PRP = 4 * DP * DP + 2;
    t.off_mean = (a.E + 1) & ~1;
    t.off_pair = t.off_mean + a.D * t.PR;
    t.CS = t.off_pair + P * t.PRP;
    t.ksep = h->sep_ks;
    t.mom_stride = (h->sep_cmax + 1) & ~1;
    t.PO = (P + a.D * (a.D + 1) + 1) & ~1;
    // Candidates per tile workgroup.  All workgroups cost the same (cch candidates + a prologue worth ~8: the tile's 128 KiB,
    // the points' inputs), 2 are resident per CU, so the launch takes ceil(workgroups / slots) rounds: pick the chunk that
    // minimises rounds x (cch + 8) -- a last round that is 16 % full cost config 4 3 % (tile_chunk 72 vs 64: 80.0 vs 78.0 ms).
    int cch = h->opt_tile_chunk;
    if (cch <= 0) {
        const long long nta = (long long)t.ntiles * a.D, slots = 2LL * h->num_cu;
        long long best = -1;
        for (int c = 16; c <= 128; c += 2) {
            const long long wgs = nta * ((a.B + c - 1) / c);
            const long long cost = ((wgs + slots - 1) / slots) * (c + 8);
            if (best < 0 || cost < best) { best = cost; cch = c; }
        }
    }
    cch = (cch + 1) & ~1;
    if (cch > 256) cch = 256;
    t.cch = cch;
    t.nchunk = (a.B + cch - 1) / cch;
}

// Whether the LDS layouts of the step kernels fit this shape (launch_rollout asks before it chooses the path).
bool tile_path_supported(Handle* h, const RolloutArgs& a) {
    if (a.D > 4) return false;
    if (ensure_sep_table(h, a.D)) return false;
    StepArgs t{};
    step_geometry(h, a, t);
    const int DP = tile_dp(a.D);
    const size_t tile_lds = (size_t)make_tile_layout(DP, a.E, t.cch).total * sizeof(double);
    const size_t point_lds = (size_t)make_point_layout(a.N, a.D, a.E, t.CS, t.mom_stride).total * sizeof(double);
    return tile_lds <= (size_t)h->lds_limit && point_lds <= (size_t)h->lds_limit;
}

// Workspace of the batch-major path: step records | per-tile partial sums | hand-over flags.
int tile_workspace(Handle* h, RolloutArgs& a) {
    int rc = ensure_sep_table(h, a.D);
    if (rc) return rc;
    StepArgs t{};
    step_geometry(h, a, t);
    const size_t nrec = (size_t)a.B * t.CS, npart = (size_t)a.B * a.D * t.ntiles, nflag = ((size_t)a.B + 1) / 2, npo = (size_t)a.B * t.PO;
    rc = grow(h, h->tilews, nrec + npart + nflag + npo);
    if (rc) return rc;
    a.tile_part = h->tilews.p + nrec;
    a.ntiles = t.ntiles;
    a.slow = reinterpret_cast<const int*>(h->tilews.p + nrec + npart);
    if (!h->side_stream) {
        // not blocking against the NULL stream: the dependencies between the two streams are the two events below
        GPMPC_HIP_CHECK(h, hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
        GPMPC_HIP_CHECK(h, hipEventCreateWithFlags(&h->ev_params, hipEventDisableTiming));
        GPMPC_HIP_CHECK(h, hipEventCreateWithFlags(&h->ev_points, hipEventDisableTiming));
    }
    return GPMPC_OK;
}

int launch_tile_state_init(Handle* h, const RolloutArgs& a, hipStream_t s) {
    const size_t n = (size_t)a.B * ((size_t)a.D + (size_t)a.D * a.D);
    hipLaunchKernelGGL(step_state_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, const_cast<int*>(a.slow));
    GPMPC_HIP_CHECK(h, hipGetLastError());
    return GPMPC_OK;
}

// One horizon step: all D x D algebra (params), the per-candidate point pass, the N x N tiles of the diagonal pairs, then the
// D x D end of the step.  Option "tile_overlap" runs the point pass on a side stream beside the tile kernel (one workgroup of
// each fits a CU together: registers and LDS).  They do run concurrently, but the step takes the same time (config 4: 77.9 ms per
// batch both ways): the tile kernel already keeps the fp64 pipe at 77 % of its nominal issue rate (2.27 GHz measured), which is
// what an 8-chain FMA loop reaches on this part (profiles/fma_loop_microbench.txt: 58.6 of 78.6 TFLOP/s), so the default is one stream.
struct TileFuse {          // gradient launch: where the fused tile pass leaves the diagonal pairs' moments
    double* tmom;          // (B, D, ntiles, kTgMom) per-tile partial moments of this step (workspace)
    double* mom;           // (B, H, P, NSP)
    int* done;             // (B, H, P)
    int NSP, NXP;
};

template <int DP>
static int launch_step_dp(Handle* h, const StepArgs& t, const TileFuse* fuse, hipStream_t s) {
    const int P = t.D * (t.D + 1) / 2;
    const bool overlap = h->opt_tile_overlap != 0;
    hipStream_t sp = overlap ? h->side_stream : s;
    hipLaunchKernelGGL(step_params_kernel<DP>, dim3((t.B * (t.D + P) + 255) / 256), dim3(256), 0, s, t);
    GPMPC_HIP_CHECK(h, hipGetLastError());
    if (overlap) {
        GPMPC_HIP_CHECK(h, hipEventRecord(h->ev_params, s));
        GPMPC_HIP_CHECK(h, hipStreamWaitEvent(sp, h->ev_params, 0));
    }
    {
        auto kern = point_pass_kernel<DP>;
        int rc = allow_full_lds(h, reinterpret_cast<const void*>(kern));
        if (rc) return rc;
        const PointLayout L = make_point_layout(t.N, t.D, t.E, t.CS, t.mom_stride);
        const size_t lds = (size_t)L.total * sizeof(double);
        if (lds > (size_t)h->lds_limit) { h->err = "point pass: LDS layout too large"; return GPMPC_ERR_LIMIT; }
        hipLaunchKernelGGL(kern, dim3(t.B), dim3(256), lds, sp, t);
        GPMPC_HIP_CHECK(h, hipGetLastError());
    }
    if (overlap) GPMPC_HIP_CHECK(h, hipEventRecord(h->ev_points, sp));
    if (fuse) {
        // gradient launch: the tile pass of this step forms the diagonal pairs' moments too (pair_tile_moments_kernel<DP, true>):
        // its W partials ARE the tile sums step_combine_kernel adds, written to t.part by the same kernel
        auto kern = pair_tile_moments_kernel<DP, true>;
        int rc = allow_full_lds(h, reinterpret_cast<const void*>(kern));
        if (rc) return rc;
        StepArgs tg = t;
        // one workgroup per CU here (2 wavefronts per SIMD): chunk = rounds x (chunk + prologue) over num_cu slots
        int cch = h->opt_tile_chunk;
        const int nta = t.ntiles * t.D;
        if (cch <= 0) {
            long long best = -1;
            for (int c = 16; c <= 128; c += 2) {
                const long long wgs = (long long)nta * ((t.B + c - 1) / c);
                const long long cost = ((wgs + h->num_cu - 1) / h->num_cu) * (c + 8);
                if (best < 0 || cost < best) { best = cost; cch = c; }
            }
        }
        cch = (cch + 1) & ~1;
        tg.cch = cch;
        tg.nchunk = (t.B + cch - 1) / cch;
        const size_t lds = (size_t)make_tile_grad_layout(DP, t.E).total * sizeof(double);
        if (lds > (size_t)h->lds_limit) { h->err = "pair tiles: input dimension too large for the LDS layout"; return GPMPC_ERR_LIMIT; }
        const int per_xcd = (nta + 7) / 8;
        hipLaunchKernelGGL(kern, dim3(8 * per_xcd * tg.nchunk), dim3(kTileWaves * 64), lds, s, tg, fuse->tmom);
        GPMPC_HIP_CHECK(h, hipGetLastError());
        hipLaunchKernelGGL(tile_moments_reduce_kernel<DP>, dim3(t.B), dim3(64), 0, s, tg, (const double*)fuse->tmom, fuse->mom, fuse->done,
                           fuse->NSP, fuse->NXP);
        GPMPC_HIP_CHECK(h, hipGetLastError());
    } else {
        auto kern = pair_tile_kernel<DP>;
        int rc = allow_full_lds(h, reinterpret_cast<const void*>(kern));
        if (rc) return rc;
        const TileLayout L = make_tile_layout(DP, t.E, t.cch);
        const size_t lds = (size_t)L.total * sizeof(double);
        if (lds > (size_t)h->lds_limit) { h->err = "pair tiles: input dimension too large for the LDS layout"; return GPMPC_ERR_LIMIT; }
        const int nta = t.ntiles * t.D;
        const int per_xcd = (nta + 7) / 8;
        hipLaunchKernelGGL(kern, dim3(8 * per_xcd * t.nchunk), dim3(kTileWaves * 64), lds, s, t);
        GPMPC_HIP_CHECK(h, hipGetLastError());
    }
    if (overlap) GPMPC_HIP_CHECK(h, hipStreamWaitEvent(s, h->ev_points, 0));
    hipLaunchKernelGGL(step_combine_kernel<DP>, dim3(t.B), dim3(64), 0, s, t);
    GPMPC_HIP_CHECK(h, hipGetLastError());
    return GPMPC_OK;
}

// Horizon step `step` of the batch-major path for every candidate the separable forms cover (a.mu_out / a.Sig_out hold the state).
int launch_pair_tiles(Handle* h, const RolloutArgs& a, int step, hipStream_t s) {
    StepArgs t{};
    step_geometry(h, a, t);
    t.Xt = a.Xt; t.beta = a.beta; t.Tm = a.Tm; t.ils2 = a.ils2; t.var = a.var; t.logvar = a.logvar; t.xrange = a.xrange;
    t.actions = a.actions;
    t.mu = a.mu_out; t.Sig = a.Sig_out;
    t.crec = h->tilews.p;
    t.part = const_cast<double*>(a.tile_part);
    t.slow = const_cast<int*>(a.slow);
    t.pout = h->tilews.p + (size_t)a.B * t.CS + (size_t)a.B * a.D * t.ntiles + ((size_t)a.B + 1) / 2;
    t.septab = h->septab; t.sepw = h->sepw.p;
    t.N = a.N; t.D = a.D; t.A = a.A; t.E = a.E; t.H = a.H; t.B = a.B; t.t = step;
    t.include_time = a.include_time; t.time0 = a.time0;
    t.force_path = a.force_path;
    t.compact = 0;
    t.fused_t = -1;
    TileFuse fz{};
    const TileFuse* fuse = nullptr;
    if (a.grad_mom) {
        // per-tile partial moments of one step's candidates (B x D x ntiles x 24 doubles: 28 MB at config 4)
        int rc = grow(h, h->tgradws, (size_t)a.B * a.D * t.ntiles * kTgMom);
        if (rc) return rc;
        fz.tmom = h->tgradws.p; fz.mom = a.grad_mom; fz.done = a.grad_done; fz.NSP = a.grad_NSP; fz.NXP = a.grad_NXP;
        fuse = &fz;
        t.fused_t = step;
    }
    switch (tile_dp(a.D)) {
        case 2:  return launch_step_dp<2>(h, t, fuse, s);
        case 3:  return launch_step_dp<3>(h, t, fuse, s);
        default: return launch_step_dp<4>(h, t, fuse, s);
    }
}

// Whether a gradient launch may leave the diagonal pairs' moments to the batch-major forward (launch_rollout decides whether the
// forward takes that path at all; h->last_fused_tiles says afterwards whether it did).
bool tile_moments_fusable(Handle* h, const RolloutArgs& a) {
    if (h->opt_grad_fuse == 0 || a.D < 2 || a.D > 4) return false;
    return (size_t)make_tile_grad_layout(tile_dp(a.D), a.E).total * sizeof(double) <= (size_t)h->lds_limit;
}

// ------------------------------------------------------------------------------------------
// Gradient: moments of the diagonal pairs of ALL (candidate, step) items of a stored trajectory, batch-major
// (pair_tile_grad_kernel.h).  Items are taken in blocks so that the records and per-tile partial moments stay a
// bounded workspace; per block: records (step_params_kernel<.., TRAJ>) -> tile moments -> ordered tile sum into the
// moment array, flag per diagonal pair for the element-wise kernels (1: written here).
template <int DP>
static int launch_tile_moments_dp(Handle* h, StepArgs t, long long items, double* mom, int* done, int NSP, int NXP, hipStream_t s) {
    auto kern = pair_tile_moments_kernel<DP>;
    int rc = allow_full_lds(h, reinterpret_cast<const void*>(kern));
    if (rc) return rc;
    const size_t lds = (size_t)make_tile_grad_layout(DP, t.E).total * sizeof(double);
    const int nta = t.ntiles * t.D;
    // items per block: workspace <= 256 MiB (config 4: ~9 000 items = ~50 rounds of workgroups per block, 7 blocks per launch;
    // the handle keeps the buffer, so its size is a standing cost next to the model)
    const size_t per_item = (size_t)t.CS + (size_t)t.D * t.ntiles * kTgMom;
    long long block = (long long)((size_t)(1u << 25) / per_item);
    if (block > items) block = items;
    if (block < 1) block = 1;
    rc = grow(h, h->tgradws, (size_t)block * per_item);
    if (rc) return rc;
    t.crec = h->tgradws.p;
    double* tmom = h->tgradws.p + (size_t)block * t.CS;
    for (long long i0 = 0; i0 < items; i0 += block) {
        const int nb = (int)((items - i0 < block) ? items - i0 : block);
        t.B = nb;
        t.item0 = (int)i0;
        // chunk: rounds x (chunk + prologue) over one workgroup per CU (2 waves per SIMD: ~200 VGPRs)
        int cch = h->opt_tile_chunk;
        if (cch <= 0) {
            long long best = -1;
            for (int c = 16; c <= 128; c += 2) {
                const long long wgs = (long long)nta * ((nb + c - 1) / c);
                const long long cost = ((wgs + h->num_cu - 1) / h->num_cu) * (c + 8);
                if (best < 0 || cost < best) { best = cost; cch = c; }
            }
        }
        cch = (cch + 1) & ~1;
        t.cch = cch;
        t.nchunk = (nb + cch - 1) / cch;
        const int P = t.D * (t.D + 1) / 2;
        hipLaunchKernelGGL((step_params_kernel<DP, true>), dim3((unsigned)(((long long)nb * (t.D + P) + 255) / 256)), dim3(256), 0, s, t);
        GPMPC_HIP_CHECK(h, hipGetLastError());
        const int per_xcd = (nta + 7) / 8;
        hipLaunchKernelGGL(kern, dim3(8 * per_xcd * t.nchunk), dim3(kTileWaves * 64), lds, s, t, tmom);
        GPMPC_HIP_CHECK(h, hipGetLastError());
        hipLaunchKernelGGL(tile_moments_reduce_kernel<DP>, dim3(nb), dim3(64), 0, s, t, (const double*)tmom, mom, done, NSP, NXP);
        GPMPC_HIP_CHECK(h, hipGetLastError());
    }
    return GPMPC_OK;
}

bool tile_moments_supported(Handle* h, const RolloutArgs& a, int NSP) {
    if (a.D < 2 || a.D > 4 || NSP > 64 || a.E - a.D > 8) return false;
    const int DP = tile_dp(a.D);
    return (size_t)make_tile_grad_layout(DP, a.E).total * sizeof(double) <= (size_t)h->lds_limit;
}

int launch_tile_moments(Handle* h, const RolloutArgs& a, double* mom, int* done, int NSP, int NXP, hipStream_t s) {
    StepArgs t{};
    step_geometry(h, a, t);
    const int DP = tile_dp(a.D);
    t.off_pair = t.off_mean;                       // compact records: inputs | the D diagonal pair problems
    t.CS = t.off_pair + a.D * t.PRP;
    t.compact = 1;
    t.fused_t = -1;
    t.Xt = a.Xt; t.beta = a.beta; t.Tm = a.Tm; t.ils2 = a.ils2; t.var = a.var; t.logvar = a.logvar; t.xrange = a.xrange;
    t.actions = a.actions;
    t.mu = a.mu_out; t.Sig = a.Sig_out;
    t.N = a.N; t.D = a.D; t.A = a.A; t.E = a.E; t.H = a.H;
    t.include_time = a.include_time; t.time0 = a.time0;
    t.force_path = a.force_path;
    const long long items = (long long)a.B * a.H;
    switch (DP) {
        case 2:  return launch_tile_moments_dp<2>(h, t, items, mom, done, NSP, NXP, s);
        case 3:  return launch_tile_moments_dp<3>(h, t, items, mom, done, NSP, NXP, s);
        default: return launch_tile_moments_dp<4>(h, t, items, mom, done, NSP, NXP, s);
    }
}

}  // namespace gpmpc_hip
