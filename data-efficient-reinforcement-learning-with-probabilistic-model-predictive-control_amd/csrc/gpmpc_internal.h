// Internal declarations shared by the HIP translation units of libgpmpc_hip.so.
// gfx950 (MI355X / CDNA4) only: 64-lane wavefronts, 160 KiB LDS per CU, 256 CUs.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdio>
#include <string>
#include <unordered_set>

#include "../../include/gpmpc.h"

namespace gpmpc_hip {

constexpr int kMaxD = GPMPC_MAX_D;
constexpr int kMaxE = GPMPC_MAX_E;
constexpr int kWave = 64;

// ---------------------------------------------------------------------------------------
// Static owners of a horizon step's work items among the CS members of a candidate (cooperative form of the fused-horizon kernel,
// rollout_kernel.h), planned once per launch on the host and read by the kernel (owner tables, LDS slots of the pairs a member
// holds records of) and by the host (sizing those slots):
//   diagonal pairs (always element-wise): their D * wtri items in order, an equal contiguous share per member;
//   off-diagonal pair of ordinal o: k = max(1, CS / P_off) consecutive members starting at o k (mod CS) share its wpp element-wise
//   slots; the one of them with the fewest diagonal items owns the pair when it is separable (a separable pair is up to 14 more
//   items, and a member with more items than wavefronts runs a second round) -- so a member holds records of at most ~2 diagonal
//   and ~1 off-diagonal pairs whatever forms the step takes (all P pairs would not fit the LDS at D = 4 or large N);
//   mean sums of output a: one member each, counted down from the last, past the owners of separable pairs (a member with a
//   separable pair AND a mean sum runs three trips of the per-point pass where the others run two).
struct ClusterMap {
    int CS, D, P, P_off, wpp, wtri, koff;
    int CSd;                            // members that own diagonal items (all of them, or all but the separable pairs' owners)
    unsigned char sepown[8];            // owner of off-diagonal pair o when it is separable (D <= 4: P_off <= 6)
    unsigned char meanown[4];           // owner of the mean sums of output a
    unsigned char dmem[32];             // the diagonal members in order, and ...
    unsigned char didx[32];             // ... a member's position among them (255: none)
    __host__ __device__ int diag_owner(int ord, int slot) const { return dmem[((ord * wtri + slot) * CSd) / (D * wtri)]; }
    __host__ __device__ int off_owner(int ord, int slot) const { return ((slot * koff) / wpp + ord * koff) % CS; }
    __host__ __device__ int sep_owner(int ord) const { return sepown[ord]; }
    __host__ __device__ int mean_owner(int a) const { return meanown[a]; }
    __host__ __device__ bool diag_needed(int ord, int m) const {
        const int i = didx[m];
        return i != 255 && ((ord * wtri) * CSd) / (D * wtri) <= i && i <= (((ord + 1) * wtri - 1) * CSd) / (D * wtri);
    }
    __host__ __device__ bool off_needed(int ord, int m) const {            // owns an element-wise slot, or the separable pair
        if (sep_owner(ord) == m) return true;
        const int mp = ((m - (ord * koff) % CS) % CS + CS) % CS;
        if (mp >= koff) return false;
        const int s0 = (mp * wpp + koff - 1) / koff;
        return s0 < wpp && (s0 * koff) / wpp == mp;
    }
    int diag_count(int m) const {                                          // element-wise items of the diagonal pairs member m owns
        const int i = didx[m], T = D * wtri;
        return i == 255 ? 0 : ((i + 1) * T + CSd - 1) / CSd - (i * T + CSd - 1) / CSd;
    }
    int slots_needed(int m) const {                                        // pairs member m holds per-point records of
        int n = 0;
        for (int a = 0; a < D; ++a) n += diag_needed(a, m) ? 1 : 0;
        for (int o = 0; o < P_off; ++o) n += off_needed(o, m) ? 1 : 0;
        return n;
    }
    void plan(int cs, int d, int wpp_, int wtri_) {                        // host
        CS = cs; D = d; P = d * (d + 1) / 2; P_off = d * (d - 1) / 2; wpp = wpp_; wtri = wtri_ > 0 ? wtri_ : 1;
        koff = P_off > 0 ? CS / P_off : 1;
        if (koff < 1) koff = 1;
        auto everybody = [&] { CSd = CS; for (int m = 0; m < 32; ++m) { dmem[m] = (unsigned char)(m < CS ? m : 0); didx[m] = (unsigned char)(m < CS ? m : 255); } };
        everybody();
        bool taken[64] = {false};
        // Members to spare (two per diagonal pair remain): the separable pairs get owners of their OWN, without diagonal items.
        // A member's per-point pass then covers two problems (a diagonal pair + a mean sum, or the two sides of its off-diagonal
        // pair) = one trip of its lanes at N <= ~220 instead of two (round 6, config 2: the three members that owned a separable
        // pair AND diagonal items had 600-800 per-point items on 448 lanes, and everybody waited for them once per step).
        const bool dedicated = P_off > 0 && P_off <= 8 && CS - P_off >= 2 * D && koff * P_off <= CS;
        for (int o = 0; o < P_off && o < 8; ++o) {
            int best = (o * koff) % CS;
            if (!dedicated)
                for (int k = 1; k < koff; ++k) {
                    const int m = (o * koff + k) % CS;
                    if (diag_count(m) <= diag_count(best)) best = m;
                }
            sepown[o] = (unsigned char)best;
            taken[best] = true;
        }
        if (dedicated) {
            CSd = 0;
            for (int m = 0; m < 32; ++m) didx[m] = 255;
            for (int m = 0; m < CS; ++m)
                if (!taken[m]) { dmem[CSd] = (unsigned char)m; didx[m] = (unsigned char)CSd; ++CSd; }
        }
        for (int a = 0; a < D && a < 4; ++a) {
            int pick = -1;
            for (int m = CS - 1, k = 0; m >= 0 && pick < 0; --m)
                if (!taken[m]) { if (k == a) pick = m; ++k; }
            meanown[a] = (unsigned char)(pick >= 0 ? pick : (a * CS / D) % CS);
        }
    }
};

// ---------------------------------------------------------------------------------------
// Kernel argument block of the rollout kernel (passed by value, < 4 KiB).
#if defined(GPMPC_HOST_TIMING)
inline double g_host_timing_fwd = 0.0;        // (timing experiment builds: when the forward launch of a gradient call was submitted)
#endif
constexpr int kInlineActs = 64;
struct RolloutArgs {
    // cached model (device)
    const double* Xt;      // (E, N)  inputs, structure-of-arrays
    const double* beta;    // (D, N)
    const double* Tm;      // (D, N, N)  beta beta^T - iK with the diagonal halved
    const double* ils2;    // (D, E)  1 / lengthscale^2
    const double* var;     // (D)     outputscale
    const double* logvar;  // (D)
    // cost block (device): target (D+A) | W (D+A)^2 | W_T D^2 | smin D | smax D
    const double* cost;
    double kappa;
    int clip;
    int use_constraints;
    // problem
    const double* actions;  // (B, H, A)
    int N, D, A, E, H, B;
    int include_time;
    double time0;
    // outputs (nullable)
    double* mu_out;
    double* Sig_out;
    double* cm_out;
    double* cv_out;
    double* J_out;
    // Taylor / separable evaluation of exp(g.w)
    const double* xrange;    // (2, E) per-input-dimension min and max of the memory points
    const int* mono_exp;     // (CM, 4) exponents of the monomials, graded order
    const double* mono_w;    // (CM) 1 / alpha!
    int mono_cum[16];        // mono_cum[k] = number of monomials of degree <= k
    int sep_kmax;            // highest degree the separable path supports (0 = disabled)
    int CM;                  // monomial capacity per (pair, side) in LDS
    int force_path;          // 0 auto, 1 always direct exp, 2 Taylor but never separable (tests)
    int force_sep;           // 1: separable whenever the degree allows, ignoring the cost model (tests)
    int x_in_lds;            // 1: X^T is copied to LDS once per launch (the per-point pass reads it every step)
    int exact_dim;           // 2: never use the compile-time-D instantiation (A/B); otherwise whenever D == DP
    int cols2;               // 1: two adjacent columns per lane in the pairwise pass (halves the LDS broadcast traffic)
    // tiling
    int G;        // output pairs per group
    int CH;       // rows per chunk
    int RC;       // row chunks per column
    unsigned magic_N;        // ceil(2^32 / NC):  x / NC  == umulhi(x, magic_N), NC = N (or ceil(N / 2) with cols2)
    unsigned magic_wpp;      // ceil(2^32 / wpp): x / wpp == umulhi(x, magic_wpp)
    unsigned magic_pt;       // ceil(2^32 / N):   x / N   == umulhi(x, magic_pt)  (per-point pass: item -> (problem, point))
    // batch-major path (pair_tile_kernel.h): the per-candidate kernel runs the horizon slice [t_begin, t_end) from the
    // state stored in mu_out / Sig_out and takes the diagonal pairs' sums from tile_part (B, D, ntiles)
    int tiled;               // host side: 1 = launch the TILED instantiation
    int t_begin, t_end;
    int ntiles;
    const double* tile_part;
    const int* slow;         // (B) t + 1: the candidate's step t is this kernel's (off-diagonal pair outside the separable range)
    // gradient launches: when the forward takes the batch-major path, its tile pass forms the moments of the diagonal pairs as
    // well (pair_tile_grad_kernel.h, FUSED) instead of being repeated by a separate moment pass over the stored trajectory
    double* grad_mom;        // (B, H, P, NSP) moment array, or NULL
    int* grad_done;          // (B, H, P) flags
    int grad_NSP, grad_NXP;
    // few-candidate cooperative form (rollout_kernel<..., CL = true>): `cluster` workgroups share one candidate's horizon step
    // and exchange their partial sums as tagged 8-byte granules through `xch` (per candidate: 2 buffers x xch_n values x 2 words)
    int cluster;                 // workgroups per candidate (1: the plain kernel)
    ClusterMap cmap;             // owners of the step's work items among the members (host-planned)
    int cl_slots;                // pairs whose per-point records a member keeps in LDS (ClusterMap::slots_needed, maximum over the members)
    int xch_n;                   // values per exchange
    unsigned xch_tag0;           // tags of this launch: xch_tag0 + step + 1 (unique over the life of the buffer)
    unsigned long long* xch;
    unsigned long long* xch_uc;  // the same layout in uncached device memory (members on several XCDs; the placement prologue)
    int defer_cost;              // 1: launch_rollout leaves the stage costs / objective to its caller (the few-candidate gradient launch folds them into its moment launch)
    int cl_dbg;                  // timing experiments of the exchange (-DGPMPC_CL_DEBUG builds only)
    // one sequence from the host (gpmpc_objective_grad_host): the actions ride in this block instead of an upload launch; the
    // fused-horizon kernel takes them from here and leaves a copy at `act_store` (= actions) for the gradient's kernels
    int act_inline_n;            // H * A (<= kInlineActs), or 0
    double* act_store;
    double act_inline[kInlineActs];
    // initial state distribution
    double mu0[kMaxD];
    double S0[kMaxD * kMaxD];
};

// A device buffer that only ever grows.
struct Buf {
    double* p = nullptr;
    size_t cap = 0;      // doubles
};

struct Handle {
    int device = 0;
    std::string err;
    // shapes of the cached model
    int N = 0, D = 0, E = 0;
    bool ready = false;
    // device buffers owned by the handle
    Buf Xt;       // (E, N)
    Buf beta;     // (D, N)
    Buf iK;       // (D, N, N)
    Buf Tm;       // (D, N, N)
    Buf ils2;     // (D, E)
    Buf var;      // (D)
    Buf logvar;   // (D)
    Buf gram;     // (D, N, N)  K + noise I, then L in its lower triangle (prepare workspace)
    Buf linv;     // (D, N, N)  L^-1 (prepare workspace)
    Buf zvec;     // (D, N)     temp for beta
    Buf cost;     // target | W | W_T | smin | smax
    Buf best;     // argmin result: [best_J, best_idx bits]
    Buf traj;     // (B, H+1, D) + (B, H+1, D, D) when the caller does not want the trajectory
    Buf xrange;   // (2, E) min / max of the inputs
    Buf gradws;   // gradient workspace: pair moments | mean sums | cost variances
    Buf mllws;    // marginal-likelihood workspace: tile partial sums | results
    Buf cemws;    // cross-entropy search workspace: optimiser vectors | model actions | J | mean | std | warm start | mapper
    Buf tilews;   // batch-major path: step records of the candidates | per-tile partial sums | hand-over flags
    Buf tgradws;  // gradient's tile pass: records of a block of (candidate, step) items | per-tile partial moments
    struct SepTable* septab = nullptr;   // monomial bands of the separable evaluation (point_pass_kernel.h), device copy
    Buf sepw;                            // their weights 1 / alpha!
    int septab_D = -1, sep_ks = 0, sep_cmax = 0;
    hipStream_t side_stream = nullptr;   // batch-major path: the point pass of a step runs beside the tile kernel
    hipEvent_t ev_params = nullptr, ev_points = nullptr;
    // incremental factorisation: what the cached factors were computed from, and border-update scratch
    Buf Xc, Yc;   // (N, E), (N, D) copies of the memory points of the last prepare
    Buf hyp;      // lengthscales (D*E) | outputscales (D) | noises (D) of the last prepare
    Buf kv, vv;   // (D, N) k(X, x_new), iK k
    Buf sc;       // (D, 2) 1 / Schur complement, v^T y
    int* mismatch = nullptr;     // device flag of the prefix comparison
    int inc_updates = 0;         // border updates since the last full factorisation
    bool have_state = false;     // Xc / Yc / hyp describe the cached factors
    Buf hio;      // gpmpc_objective_grad_host: actions | J | grad | mu | Sig | cost_mu | cost_var of one candidate (device side)
    double* hio_host = nullptr;      // ... and its pinned, device-mapped host mirror (results)
    double* hio_host_dev = nullptr;
    size_t hio_host_cap = 0;
    // ... whose results the reverse sweep itself copies to the host mirror before it raises a sequence number there (the host
    // polls that word instead of synchronising the stream): set for the one launch of gpmpc_objective_grad_host
    unsigned long long* hio_flag = nullptr;          // pinned host word and its device address
    unsigned long long* hio_flag_dev = nullptr;
    unsigned long long hio_seq = 0;
    double* hx_out = nullptr; const double* hx_src = nullptr; int hx_n = 0;      // export request (hx_n = 0: none)
    Buf xch;      // exchange granules of the cooperative few-candidate kernel (zeroed when (re)allocated, tags never repeat)
    unsigned long long* xch_uc = nullptr;    // ... its uncached twin (hipExtMallocWithFlags)
    size_t xch_uc_cap = 0;       // 8-byte words
    unsigned xch_epoch = 1;      // launches that used the exchange buffers, from 1: tag 0 is what a zeroed buffer holds and must never be
                                 // a tag anybody waits for (a fresh handle's first launch accepted the zeros as the members' XCD ids)
    int opt_cluster = 0;         // workgroups per candidate of the few-candidate path: 0 auto, 1 never, n >= 2 fixed
    int opt_cl_dbg = 0;
    int last_cluster = 1;        // what the last fused-horizon launch used
    Buf mono_w;   // (CM) 1 / alpha!
    int* mono_exp = nullptr;    // (CM, 4)
    int mono_D = -1;            // state dimension the monomial tables were built for
    int mono_cum[16] = {0};
    int sep_kmax = 0;
    int mono_CM = 0;
    int* info = nullptr;        // (kMaxD) first non-positive pivot + 1, or 0
    std::unordered_set<const void*> lds_configured;   // kernels whose dynamic-LDS limit was raised on this device
    // cost settings
    int cost_D = -1, cost_A = -1;
    double kappa = 1.0;
    int clip = 0;
    int use_constraints = 0;
    // options
    int opt_threads = 0;
    int opt_lds_kb = 0;              // > 0: LDS budget (KiB) of a fused-horizon workgroup (with threads = 512: two workgroups per CU)
    int opt_force_global = 0;
    int opt_rows_per_chunk = 0;
    int opt_force_path = 0;
    int opt_force_sep = 0;
    int opt_grad_stream = 0;         // 1: always the streaming moment pass of the gradient (tests); otherwise only when N needs it
    int opt_grad_share = 0;          // moment pass of the gradient, D <= 3: two 512-thread workgroups per CU -- 0 auto (grad.hip), 1 where the LDS fits twice, 2 never
    int opt_grad_chunk = 0;          // rows per work item of the LDS-resident moment pass: 0 = chosen by the schedule model (grad.hip), else fixed (multiple of 4, <= 64)
    int chunk_key[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // the model's last answer: (N, D, E, columns per lane, waves, LDS KiB, small-batch items, pairs left) -> chunk_rows
    int chunk_rows = 0;
    int opt_grad_tiles = 1;          // diagonal pairs of the gradient's moment pass batch-major (pair_tile_grad_kernel.h): 0 never, 1 auto, 2 always
    int opt_grad_sep = 1;            // off-diagonal pairs of the gradient's moment pass in separable form on the matrix cores:
                                     // 0 never (element-wise), 1 from N = 128 on when B x H fills the chip (measured crossover, grad.hip), 2 always (tests, A/B)
    int opt_cols_per_lane = 0;       // 0 auto, 1 / 2: columns per lane in the pairwise pass of the rollout kernel
    int opt_incremental = 1;         // reuse / border-update the cached factors when the memory only grew
    int opt_refresh_every = 32;      // full refactorisation after this many border updates (bounds drift)
    int opt_outer2 = 2;              // large N: binary outer levels of the trailing update, up to 128 * 2^value columns (0: 128 only)
    int opt_outer_min_n = 640;       // memories from this size on take the outer-panel factorisation path (measured crossover:
                                     // N = 500: 0.61 vs 0.66 ms, 640: 0.92 vs 0.92, 768: 1.22 vs 1.07, 1000: 1.81 vs 1.37; tools/gpu_outer_min_n.py)
    int opt_block128 = 1;            // large N: a whole 128-column outer panel in two launches (LDS-resident block factorisation + tiled solve)
    int opt_inner_left = 1;          // large N: left-looking 32-column panels inside an outer panel; 0: right-looking (A/B)
    int opt_tile128 = 1;             // large N: 128 x 128 tiles (8 wavefronts) for the tiled products; 0: 64 x 64 (A/B)
    int opt_outer_block = 1;         // large N: outer panels of 128 columns + LDS-tiled products; 0: the 32-wide path only (A/B, tests)
    int opt_pair_tiles = 0;          // batch-major pairwise pass of the diagonal pairs: 0 auto (by N, D, B), 1 always (D <= 4), 2 never
    int opt_tile_chunk = 0;          // candidates per workgroup of the batch-major pass (0: chosen from the batch)
    int opt_tile_overlap = 0;        // batch-major path: 1 = point pass on a side stream, concurrent with the tile kernel.  Measured at config 4
                                     // (round 3): 77.9 ms either way -- the two kernels do overlap (rocprofv3: 2.58 ms and 1.41 ms side by side instead
                                     // of 2.14 + 0.49 ms) but the fp64 pipe is already at the ~76 % of its nominal rate an FMA loop reaches
    int last_rollout_path = 0;       // what the last rollout launch used: 0 fused-horizon kernel, 1 streaming kernel, 2 batch-major tiles
    int last_fused_tiles = 0;        // 1: the last batch-major forward also formed the gradient's tile moments (RolloutArgs::grad_mom)
    int opt_grad_merge = 1;          // few candidates: element-wise moments + mean moments + stage costs in ONE launch (0: three launches, A/B)
    int opt_grad_mean = 1;           // moment pass (D <= 4): the mean part by mean_moments_kernel (lanes over points); 0: inside the pass (A/B, tests)
    int opt_grad_fuse = 1;           // gradient: form the diagonal pairs' tile moments inside the batch-major forward (0: separate pass, A/B)
    int last_grad_path = 0;          // moment passes of the last gpmpc_rollout_grad: bit 0 separable off-diagonal pairs, bit 1 tile moments of
                                     // the diagonal pairs, bit 2 streaming element-wise pass, bit 3 the wide (8 < D <= 16) pass
    int opt_prepare_inv_batch = 4;   // 32-wide panel path: row blocks of L^-1 per side-stream launch (1: one launch per row block)
    int opt_prepare_invcols = 1;     // 32-wide panel path, N <= 352 (measured crossover; option 2: up to 544): L^-1 in one launch after the factorisation (0: row blocks beside / after the panels)
    int opt_prepare_fuse = 1;        // 32-wide panel path: trailing update fused with the next diagonal block's factorisation (0: separate launches, A/B)
    int opt_prepare_overlap = 1;     // 32-wide panel path: the inverse's launches on a side stream beside the factorisation's (0: one stream, A/B)
    int opt_gram_shared = 1;         // large N: K of all GPs by one workgroup per tile, squared differences shared (0: per-GP kernel, A/B)
    int opt_fused_prepare = 1;       // N <= 256: the whole factorisation in one launch (prepare_small.hip); 0: panel path (A/B, tests)
    int last_prepare_mode = 0;       // 0 full, 1 border update(s), 2 unchanged (cache hit)
    int lds_limit = 160 * 1024;
    int num_cu = 256;
};

// Raise a kernel's dynamic-LDS limit to the device maximum once per handle (= per device).
inline int allow_full_lds(Handle* h, const void* kernel);

#define GPMPC_HIP_CHECK(h, expr)                                                         \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                \
            return GPMPC_ERR_HIP;                                                        \
        }                                                                                \
    } while (0)

inline int allow_full_lds(Handle* h, const void* kernel) {
    if (h->lds_configured.count(kernel)) return GPMPC_OK;
    GPMPC_HIP_CHECK(h, hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_limit));
    h->lds_configured.insert(kernel);
    return GPMPC_OK;
}

// rollout.hip
int launch_rollout(Handle* h, RolloutArgs& a, hipStream_t s);
int launch_argmin(Handle* h, const double* J, int B, long long first, hipStream_t s);
int launch_traj_cost(Handle* h, const RolloutArgs& a, double* cm, double* cv, double* J, hipStream_t s);     // stage costs + objective of the stored trajectory
// pair_tile.hip: the batch-major pass of horizon step t (a.mu_out / a.Sig_out hold the state), a.tile_part / a.ntiles set on return
bool tile_path_supported(Handle* h, const RolloutArgs& a);
int tile_workspace(Handle* h, RolloutArgs& a);
int launch_tile_state_init(Handle* h, const RolloutArgs& a, hipStream_t s);
int launch_pair_tiles(Handle* h, const RolloutArgs& a, int t, hipStream_t s);
bool tile_moments_fusable(Handle* h, const RolloutArgs& a);
bool tile_moments_supported(Handle* h, const RolloutArgs& a, int NSP);
int launch_tile_moments(Handle* h, const RolloutArgs& a, double* mom, int* done, int NSP, int NXP, hipStream_t s);
// grad.hip
int launch_rollout_grad(Handle* h, RolloutArgs& a, double* grad_out, hipStream_t s);
int launch_rollout_grad_wide(Handle* h, RolloutArgs& a, double* grad_out, hipStream_t s);     // grad_wide.hip: 8 < D <= 16
int launch_argmin_to(Handle* h, const double* J, int B, long long first, const double* actions, int HA, double* out_dev,
                     hipStream_t s);
// prepare.hip
int run_prepare(Handle* h, const double* X, const double* Y, const double* ls, const double* os,
                const double* noise, int N, int D, int E, hipStream_t s);
int run_set_factors(Handle* h, const double* X, const double* iK, const double* beta,
                    const double* ls, const double* os, int N, int D, int E, hipStream_t s);
int ensure_model_buffers(Handle* h, int N, int D, int E, bool need_factor_ws);
int run_mll(Handle* h, const double* X, const double* Y, const double* ls, const double* os, const double* noise,
            int N, int D, int E, double* out_host, hipStream_t s);
int grow(Handle* h, Buf& b, size_t need);
// search.hip: cross-entropy search whose loop stays on the device (actions / J_out of `a` are set inside)
int run_cem_search(Handle* h, RolloutArgs& a, int iterations, int n_elite, unsigned long long seed, const double* first_host,
                   int mapper, const double* max_change_host, const double* a_prev_host, const double* noise_dev,
                   double* best_out_dev, hipStream_t s);
// ... and its per-iteration halves for candidates sharded over GPUs (a.B = the slice length)
int run_cem_local(Handle* h, RolloutArgs& a, int B_total, int b0, int it, int n_elite, unsigned long long seed,
                  const double* first_host, int mapper, const double* max_change_host, const double* a_prev_host,
                  const double* noise_dev, const double* state_dev, double* elites_out_dev, hipStream_t s);
int run_cem_merge(Handle* h, const double* elites_dev, int lists, int n_elite, int n, int it, double* state_dev, hipStream_t s);
// prepare_small.hip: 1 = handled (N <= 256), 0 = not applicable, < 0 = error
int run_prepare_small(Handle* h, const double* X, const double* Y, const double* ls, const double* os, const double* noise,
                      int N, int D, int E, hipStream_t s);

}  // namespace gpmpc_hip
