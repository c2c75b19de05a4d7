// pair_tile.hip -- host side of the batch-major pairwise pass (pair_tile_kernel.h): workspace, tiling, launches.
#include "pair_tile_kernel.h"

namespace gpmpc_hip {

static int tile_dp(int D) { return D <= 2 ? 2 : (D == 3 ? 3 : 4); }

static void tile_geometry(const Handle* h, const RolloutArgs& a, TileArgs& t) {
    const int DP = tile_dp(a.D);
    t.nb = (a.N + kTileW - 1) / kTileW;
    t.ntiles = t.nb * (t.nb + 1) / 2;
    t.PS = tile_par_stride(DP, a.E);
    // candidates per workgroup: enough workgroups for ~8 rounds over the 2 x 256 resident ones (tail), few enough that the
    // tile's 128 KiB and the prologue are amortised over >= 16 candidates; even (records are made two candidates at a time)
    int cch = h->opt_tile_chunk;
    if (cch <= 0) {
        const long long nta = (long long)t.ntiles * a.D;
        cch = (int)(((long long)a.B * nta + 4095) / 4096);
        if (cch < 16) cch = 16;
        if (cch > 128) cch = 128;
    }
    cch = (cch + 1) & ~1;
    if (cch > 256) cch = 256;
    t.cch = cch;
    t.nchunk = (a.B + cch - 1) / cch;
}

// Workspace of the batch-major path: per-(candidate, output) parameters | per-tile partial sums.
int tile_workspace(Handle* h, RolloutArgs& a) {
    TileArgs t{};
    tile_geometry(h, a, t);
    const size_t npar = (size_t)a.B * a.D * t.PS, npart = (size_t)a.B * a.D * t.ntiles;
    int rc = grow(h, h->tilews, npar + npart);
    if (rc) return rc;
    a.tile_part = h->tilews.p + npar;
    a.ntiles = t.ntiles;
    return GPMPC_OK;
}

int launch_tile_state_init(Handle* h, const RolloutArgs& a, hipStream_t s) {
    const size_t n = (size_t)a.B * ((size_t)a.D + (size_t)a.D * a.D);
    hipLaunchKernelGGL(tile_state_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    GPMPC_HIP_CHECK(h, hipGetLastError());
    return GPMPC_OK;
}

template <int DP>
static int launch_tiles_dp(Handle* h, const TileArgs& t, hipStream_t s) {
    hipLaunchKernelGGL(tile_params_kernel<DP>, dim3((t.B * t.D + 255) / 256), dim3(256), 0, s, t);
    GPMPC_HIP_CHECK(h, hipGetLastError());
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
    return GPMPC_OK;
}

int launch_pair_tiles(Handle* h, const RolloutArgs& a, int step, hipStream_t s) {
    TileArgs t{};
    tile_geometry(h, a, t);
    t.Xt = a.Xt; t.Tm = a.Tm; t.ils2 = a.ils2; t.logvar = a.logvar; t.xrange = a.xrange; t.actions = a.actions;
    t.mu = a.mu_out; t.Sig = a.Sig_out;
    t.tpar = h->tilews.p;
    t.part = const_cast<double*>(a.tile_part);
    t.N = a.N; t.D = a.D; t.A = a.A; t.E = a.E; t.H = a.H; t.B = a.B; t.t = step;
    t.include_time = a.include_time; t.time0 = a.time0;
    t.force_path = a.force_path;
    switch (tile_dp(a.D)) {
        case 2:  return launch_tiles_dp<2>(h, t, s);
        case 3:  return launch_tiles_dp<3>(h, t, s);
        default: return launch_tiles_dp<4>(h, t, s);
    }
}

}  // namespace gpmpc_hip
