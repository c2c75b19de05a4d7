// rollout.hip -- host-side dispatch of the rollout kernel (tiling choice, LDS sizing) and the
// argmin kernel.  The kernel itself lives in rollout_kernel.h / rollout_dp.hip.
#include "rollout_kernel.h"

namespace gpmpc_hip {

int launch_rollout_dp2(Handle*, RolloutArgs&, int, bool, size_t, hipStream_t);
int launch_rollout_dp3(Handle*, RolloutArgs&, int, bool, size_t, hipStream_t);
int launch_rollout_dp4(Handle*, RolloutArgs&, int, bool, size_t, hipStream_t);
int launch_rollout_dp6(Handle*, RolloutArgs&, int, bool, size_t, hipStream_t);
int launch_rollout_dp8(Handle*, RolloutArgs&, int, bool, size_t, hipStream_t);
int launch_rollout_dp16(Handle*, RolloutArgs&, int, bool, size_t, hipStream_t);

// ------------------------------------------------------------------------------------------
// Keep-the-best rule of gp_mpc_controller.py:146-148 over a vector (single workgroup).
__global__ __launch_bounds__(1024) void argmin_kernel(const double* J, int B, long long first, double* out) {
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
        reinterpret_cast<long long*>(out)[1] = (bi < 0) ? -1 : bi + first;
    }
}

int launch_argmin(Handle* h, const double* J, int B, long long first, hipStream_t s) {
    int rc = grow(h, h->best, 2);
    if (rc) return rc;
    hipLaunchKernelGGL(argmin_kernel, dim3(1), dim3(1024), 0, s, J, B, first, h->best.p);
    GPMPC_HIP_CHECK(h, hipGetLastError());
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

    // workgroup size: one candidate per workgroup; 16 waves when the batch does not
    // oversubscribe the 256 CUs, fewer (more workgroups per CU) when it does.
    int nt = h->opt_threads;
    if (nt != 256 && nt != 512 && nt != 1024) {
        if (a.B <= h->num_cu) nt = 1024;
        else if (a.B <= 2 * h->num_cu) nt = 512;
        else nt = (N >= 128) ? 512 : 256;
    }
    const int nw = nt / 64;

    // choose pairs-per-group G and row chunking for the LDS-resident variant
    bool gs = h->opt_force_global != 0;
    int G = 0, CH = 0, RC = 0;
    size_t lds_bytes = 0;
    auto chunking = [&](int g) {
        // aim for ~8 wave items per wave and group, chunks of at least 16 rows
        long long want = 8LL * nw * 64;
        long long rc = (want + (long long)g * N - 1) / ((long long)g * N);
        int maxrc = (N + 15) / 16;
        if (rc > maxrc) rc = maxrc;
        if (rc < 1) rc = 1;
        if (h->opt_rows_per_chunk > 0) rc = (N + h->opt_rows_per_chunk - 1) / h->opt_rows_per_chunk;
        CH = (N + (int)rc - 1) / (int)rc;
        RC = (N + CH - 1) / CH;
    };
    if (!gs) {
        for (int g = P; g >= 1; --g) {
            chunking(g);
            const int wpp = (RC * N + 63) / 64;
            Layout L = make_layout(N, D, A, E, g, DP, wpp, false);
            if ((size_t)L.lds_total * 8 <= (size_t)h->lds_limit) { G = g; lds_bytes = (size_t)L.lds_total * 8; break; }
        }
        if (G == 0) gs = true;
    }
    if (gs) {
        // large-N variant: per-point arrays in per-candidate global scratch (L2 resident),
        // small algebra in LDS
        for (int g = (P < 16 ? P : 16); g >= 1; --g) {
            chunking(g);
            const int wpp = (RC * N + 63) / 64;
            Layout L = make_layout(N, D, A, E, g, DP, wpp, true);
            if ((size_t)L.lds_total * 8 <= (size_t)h->lds_limit) {
                G = g; lds_bytes = (size_t)L.lds_total * 8;
                const size_t need = (size_t)L.pp_total * a.B;
                int rc = grow(h, h->scratch, need);
                if (rc) return rc;
                a.scratch = h->scratch.p;
                a.scratch_stride = (size_t)L.pp_total;
                break;
            }
        }
        if (G == 0) { h->err = "rollout: problem does not fit LDS even with global scratch"; return GPMPC_ERR_LIMIT; }
    }
    a.G = G; a.CH = CH; a.RC = RC;

    switch (DP) {
        case 2:  return launch_rollout_dp2(h, a, nt, gs, lds_bytes, s);
        case 3:  return launch_rollout_dp3(h, a, nt, gs, lds_bytes, s);
        case 4:  return launch_rollout_dp4(h, a, nt, gs, lds_bytes, s);
        case 6:  return launch_rollout_dp6(h, a, nt, gs, lds_bytes, s);
        case 8:  return launch_rollout_dp8(h, a, nt, gs, lds_bytes, s);
        default: return launch_rollout_dp16(h, a, nt, gs, lds_bytes, s);
    }
}

}  // namespace gpmpc_hip
