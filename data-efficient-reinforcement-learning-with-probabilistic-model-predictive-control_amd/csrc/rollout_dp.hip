// rollout_dp.hip -- one translation unit per padded state dimension (compiled with
// -DGPMPC_DP=<n>) so that the template instantiations build in parallel.
#include "rollout_stream_kernel.h"

#ifndef GPMPC_DP
#error "compile with -DGPMPC_DP=<padded state dimension>"
#endif

namespace gpmpc_hip {

// ------------------------------------------------------------------------------------------
template <int DP, int NT>
static int launch_variant(Handle* h, RolloutArgs& a, bool global_scratch, size_t lds_bytes, hipStream_t s) {
    if (global_scratch) {
        // large-N variant: nothing per-point is materialised (rollout_stream_kernel.h); always 1024 threads
        auto sk = rollout_stream_kernel<DP, 1024>;
        int rc = allow_full_lds(h, reinterpret_cast<const void*>(sk));
        if (rc) return rc;
        hipLaunchKernelGGL(sk, dim3(a.B), dim3(1024), lds_bytes, s, a);
        GPMPC_HIP_CHECK(h, hipGetLastError());
        return GPMPC_OK;
    }
    // The compile-time-D instantiation folds the index arithmetic.  For D >= 4 it spills (8 / 22 / 247 / 1000 VGPRs at
    // D = 4 / 6 / 8 / 16 against none with the runtime D) but only in the small algebra outside the pairwise loop, and it
    // still measured faster or equal at every D (tools/gpu_exact_ab.py: 19.1 vs 19.8 ms at config 4, 5.1 vs 6.0 ms at
    // D = 6, 14.2 vs 15.6 ms at D = 8, equal at D = 16), so it is used whenever D == DP; option "exact_dim" = 2 forbids it.
    const bool exact = (a.D == DP) && a.exact_dim != 2;
    if constexpr (DP <= 4 && NT == 1024) {
        if (a.tiled) {
            // one horizon step of the per-candidate part of the batch-major path (diagonal pairs: pair_tile_kernel.h)
            auto tk = exact ? rollout_kernel<DP, NT, DP, true, true> : rollout_kernel<DP, NT, 0, true, true>;
            int rc = allow_full_lds(h, reinterpret_cast<const void*>(tk));
            if (rc) return rc;
            hipLaunchKernelGGL(tk, dim3(a.B), dim3(NT), lds_bytes, s, a);
            GPMPC_HIP_CHECK(h, hipGetLastError());
            return GPMPC_OK;
        }
    }
    if (a.tiled) { h->err = "rollout: batch-major path asked for an unsupported kernel variant"; return GPMPC_ERR_ARG; }
    if (a.cluster > 1) {
        // few-candidate cooperative form: a.cluster workgroups per candidate, candidates in groups of 8 (one per XCD)
        if constexpr (DP <= 4) {
            if (!a.cols2) { h->err = "rollout: the cooperative form needs the two-column pair pass"; return GPMPC_ERR_ARG; }
            auto ck = exact ? rollout_kernel<DP, NT, DP, true, false, true> : rollout_kernel<DP, NT, 0, true, false, true>;
            int rc = allow_full_lds(h, reinterpret_cast<const void*>(ck));
            if (rc) return rc;
            hipLaunchKernelGGL(ck, dim3(8 * a.cluster * ((a.B + 7) / 8)), dim3(NT), lds_bytes, s, a);
            GPMPC_HIP_CHECK(h, hipGetLastError());
            return GPMPC_OK;
        } else {
            h->err = "rollout: cooperative form asked for an unsupported kernel variant"; return GPMPC_ERR_ARG;
        }
    }
    auto kern = a.cols2 ? (exact ? rollout_kernel<DP, NT, DP, true> : rollout_kernel<DP, NT, 0, true>)
                        : (exact ? rollout_kernel<DP, NT, DP, false> : rollout_kernel<DP, NT, 0, false>);
    {
        int rc = allow_full_lds(h, reinterpret_cast<const void*>(kern));
        if (rc) return rc;
    }
    hipLaunchKernelGGL(kern, dim3(a.B), dim3(NT), lds_bytes, s, a);
    GPMPC_HIP_CHECK(h, hipGetLastError());
    return GPMPC_OK;
}

template <int DP>
static int launch_dp(Handle* h, RolloutArgs& a, int nt, bool gs, size_t lds, hipStream_t s) {
    switch (nt) {
        case 256: return launch_variant<DP, 256>(h, a, gs, lds, s);
        case 512: return launch_variant<DP, 512>(h, a, gs, lds, s);
        default:  return launch_variant<DP, 1024>(h, a, gs, lds, s);
    }
}


#define GPMPC_CAT_(a, b) a##b
#define GPMPC_CAT(a, b) GPMPC_CAT_(a, b)
int GPMPC_CAT(launch_rollout_dp, GPMPC_DP)(Handle* h, RolloutArgs& a, int nt, bool gs, size_t lds, hipStream_t s) {
    return launch_dp<GPMPC_DP>(h, a, nt, gs, lds, s);
}

}  // namespace gpmpc_hip
