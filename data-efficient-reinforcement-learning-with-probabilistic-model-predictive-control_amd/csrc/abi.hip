// abi.hip -- extern "C" entry points of libgpmpc_hip.so (declared in include/gpmpc.h).
#include <cstring>
#include <new>
#include <dlfcn.h>

#include "gpmpc_internal.h"

// ROCTx ranges around the entry points (tracing row): visible in `rocprofv3 --marker-trace` timelines
// as gpmpc_prepare / gpmpc_rollout / gpmpc_rollout_grad / gpmpc_argmin.  The marker library is looked up at first use
// (rocprofiler-sdk's roctx, then the older roctracer one); without it the ranges cost one predictable branch.
namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        for (const char* lib : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
            void* hnd = dlopen(lib, RTLD_LAZY | RTLD_LOCAL);
            if (!hnd) continue;
            push = reinterpret_cast<int (*)(const char*)>(dlsym(hnd, "roctxRangePushA"));
            pop = reinterpret_cast<int (*)()>(dlsym(hnd, "roctxRangePop"));
            if (push && pop) return;
            push = nullptr; pop = nullptr;
        }
    }
};
struct Range {
    static const Roctx& api() { static Roctx r; return r; }
    explicit Range(const char* name) { if (api().push) api().push(name); }
    ~Range() { if (api().pop) api().pop(); }
    Range(const Range&) = delete;
    Range& operator=(const Range&) = delete;
};
}  // namespace

using namespace gpmpc_hip;

struct gpmpc {
    Handle h;
};

#define H_(x) (&(x)->h)

static int bad(gpmpc_t* g, const char* msg) {
    if (g) g->h.err = msg;
    return GPMPC_ERR_ARG;
}

static void free_buf(Buf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

extern "C" {

int gpmpc_abi_version(void) { return 12; }

int gpmpc_create(gpmpc_t** out, int device_id) {
    if (!out) return GPMPC_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return GPMPC_ERR_HIP;
    if (hipSetDevice(device_id) != hipSuccess) return GPMPC_ERR_HIP;
    gpmpc_t* g = new (std::nothrow) gpmpc();
    if (!g) return GPMPC_ERR_HIP;
    g->h.device = device_id;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) {
        g->h.num_cu = prop.multiProcessorCount;
        if (prop.maxSharedMemoryPerMultiProcessor >= 64 * 1024)
            g->h.lds_limit = (int)prop.maxSharedMemoryPerMultiProcessor;
        if (g->h.lds_limit > 160 * 1024) g->h.lds_limit = 160 * 1024;
    }
    *out = g;
    return GPMPC_OK;
}

int gpmpc_destroy(gpmpc_t* g) {
    if (!g) return GPMPC_ERR_ARG;
    Handle* h = H_(g);
    (void)hipSetDevice(h->device);
    Buf* all[] = {&h->Xt, &h->beta, &h->iK, &h->Tm, &h->ils2, &h->var, &h->logvar, &h->gram,
                  &h->linv, &h->zvec, &h->cost, &h->best, &h->xrange, &h->mono_w, &h->traj, &h->Xc, &h->Yc,
                  &h->hyp, &h->kv, &h->vv, &h->sc, &h->gradws, &h->mllws, &h->cemws, &h->tilews, &h->sepw, &h->tgradws, &h->xch, &h->hio};
    for (Buf* b : all) free_buf(*b);
    if (h->hio_host) (void)hipHostFree(h->hio_host);
    if (h->hio_flag) (void)hipHostFree(h->hio_flag);
    if (h->xch_uc) (void)hipFree(h->xch_uc);
    if (h->info) (void)hipFree(h->info);
    if (h->mono_exp) (void)hipFree(h->mono_exp);
    if (h->septab) (void)hipFree(h->septab);
    if (h->side_stream) (void)hipStreamDestroy(h->side_stream);
    if (h->ev_params) (void)hipEventDestroy(h->ev_params);
    if (h->ev_points) (void)hipEventDestroy(h->ev_points);
    delete g;
    return GPMPC_OK;
}

const char* gpmpc_last_error(const gpmpc_t* g) { return g ? g->h.err.c_str() : "null handle"; }

int gpmpc_set_option(gpmpc_t* g, const char* name, long long value) {
    if (!g || !name) return GPMPC_ERR_ARG;
    Handle* h = H_(g);
    if (!strcmp(name, "threads")) h->opt_threads = (int)value;
    else if (!strcmp(name, "lds_limit_kb")) h->opt_lds_kb = (int)value;
    else if (!strcmp(name, "force_global_scratch")) h->opt_force_global = (int)value;
    else if (!strcmp(name, "rows_per_chunk")) h->opt_rows_per_chunk = (int)value;
    else if (!strcmp(name, "force_path")) h->opt_force_path = (int)value;
    else if (!strcmp(name, "force_separable")) h->opt_force_sep = (int)value;
    else if (!strcmp(name, "cols_per_lane")) h->opt_cols_per_lane = (int)value;
    else if (!strcmp(name, "grad_share_cu")) h->opt_grad_share = (int)value;
    else if (!strcmp(name, "grad_chunk_rows")) {
        const int v = (int)value;
        if (v < 0 || v > 64 || (v & 3)) { h->err = "grad_chunk_rows: 0 (auto) or a multiple of 4 up to 64"; return GPMPC_ERR_ARG; }
        h->opt_grad_chunk = v;
    }
    else if (!strcmp(name, "incremental")) h->opt_incremental = (int)value;
    else if (!strcmp(name, "grad_stream")) h->opt_grad_stream = (int)value;
    else if (!strcmp(name, "grad_separable")) h->opt_grad_sep = (int)value;
    else if (!strcmp(name, "grad_tiles")) h->opt_grad_tiles = (int)value;
    else if (!strcmp(name, "grad_fuse")) h->opt_grad_fuse = (int)value;
    else if (!strcmp(name, "grad_mean")) h->opt_grad_mean = (int)value;
    else if (!strcmp(name, "grad_merge")) h->opt_grad_merge = (int)value;
    else if (!strcmp(name, "prepare_overlap")) h->opt_prepare_overlap = (int)value;
    else if (!strcmp(name, "prepare_fuse")) h->opt_prepare_fuse = (int)value;
    else if (!strcmp(name, "prepare_invcols")) h->opt_prepare_invcols = (int)value;
    else if (!strcmp(name, "prepare_inv_batch")) h->opt_prepare_inv_batch = (int)value;
    else if (!strcmp(name, "fused_prepare")) h->opt_fused_prepare = (int)value;
    else if (!strcmp(name, "gram_shared")) h->opt_gram_shared = (int)value;
    else if (!strcmp(name, "outer_block")) h->opt_outer_block = (int)value;
    else if (!strcmp(name, "tile128")) h->opt_tile128 = (int)value;
    else if (!strcmp(name, "block128")) h->opt_block128 = (int)value;
    else if (!strcmp(name, "outer_min_n")) h->opt_outer_min_n = (int)value < 256 ? 256 : (int)value;
    else if (!strcmp(name, "inner_left")) h->opt_inner_left = (int)value;
    else if (!strcmp(name, "outer2")) h->opt_outer2 = (int)value;
    else if (!strcmp(name, "refresh_every")) h->opt_refresh_every = (int)value;
    else if (!strcmp(name, "pair_tiles")) h->opt_pair_tiles = (int)value;
    else if (!strcmp(name, "tile_chunk")) h->opt_tile_chunk = (int)value;
    else if (!strcmp(name, "tile_overlap")) h->opt_tile_overlap = (int)value;
    else if (!strcmp(name, "cluster_debug")) h->opt_cl_dbg = (int)value;
    else if (!strcmp(name, "cluster")) {
        if (value < 0 || value > 32) { h->err = "cluster: 0 (auto), 1 (never) or 2..32 workgroups per candidate"; return GPMPC_ERR_ARG; }
        h->opt_cluster = (int)value;
    }
    else return bad(g, "unknown option");
    return GPMPC_OK;
}

static int check_dims(gpmpc_t* g, int N, int D, int E) {
    if (N < 1 || D < 1 || E < D) return bad(g, "need N >= 1, D >= 1, E >= D");
    if (D > kMaxD || E > kMaxE) { g->h.err = "shape outside compiled limits (D <= 16, E <= 24)"; return GPMPC_ERR_LIMIT; }
    return GPMPC_OK;
}

int gpmpc_prepare(gpmpc_t* g, const double* X, const double* Y, const double* ls, const double* os,
                  const double* noise, int N, int D, int E, void* stream) {
    Range roctx_range("gpmpc_prepare");
    if (!g || !X || !Y || !ls || !os || !noise) return bad(g, "null argument");
    int rc = check_dims(g, N, D, E);
    if (rc) return rc;
    GPMPC_HIP_CHECK(H_(g), hipSetDevice(g->h.device));
    return run_prepare(H_(g), X, Y, ls, os, noise, N, D, E, (hipStream_t)stream);
}

int gpmpc_set_factors(gpmpc_t* g, const double* X, const double* iK, const double* beta, const double* ls,
                      const double* os, int N, int D, int E, void* stream) {
    Range roctx_range("gpmpc_set_factors");
    if (!g || !X || !iK || !beta || !ls || !os) return bad(g, "null argument");
    int rc = check_dims(g, N, D, E);
    if (rc) return rc;
    GPMPC_HIP_CHECK(H_(g), hipSetDevice(g->h.device));
    return run_set_factors(H_(g), X, iK, beta, ls, os, N, D, E, (hipStream_t)stream);
}

int gpmpc_get_factors(gpmpc_t* g, const double** iK, const double** beta) {
    if (!g || !g->h.ready) return bad(g, "no factors cached: call gpmpc_prepare first");
    if (iK) *iK = g->h.iK.p;
    if (beta) *beta = g->h.beta.p;
    return GPMPC_OK;
}

int gpmpc_read_factors(gpmpc_t* g, double* iK_dst, double* beta_dst, void* stream) {
    if (!g || !g->h.ready) return bad(g, "no factors cached: call gpmpc_prepare first");
    Handle* h = H_(g);
    GPMPC_HIP_CHECK(h, hipSetDevice(h->device));
    const size_t DN = (size_t)h->D * h->N;
    if (iK_dst) GPMPC_HIP_CHECK(h, hipMemcpyAsync(iK_dst, h->iK.p, DN * h->N * sizeof(double), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    if (beta_dst) GPMPC_HIP_CHECK(h, hipMemcpyAsync(beta_dst, h->beta.p, DN * sizeof(double), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return GPMPC_OK;
}

int gpmpc_mll(gpmpc_t* g, const double* X, const double* Y, const double* ls, const double* os, const double* noise,
              int N, int D, int E, double* out_host, void* stream) {
    Range roctx_range("gpmpc_mll");
    if (!g) return GPMPC_ERR_ARG;
    if (!X || !Y || !ls || !os || !noise || !out_host) return bad(g, "null argument");
    int rc = check_dims(g, N, D, E);
    if (rc) return rc;
    GPMPC_HIP_CHECK(H_(g), hipSetDevice(g->h.device));
    return run_mll(H_(g), X, Y, ls, os, noise, N, D, E, out_host, (hipStream_t)stream);
}

int gpmpc_last_prepare_mode(gpmpc_t* g) { return g ? g->h.last_prepare_mode : GPMPC_ERR_ARG; }

int gpmpc_last_rollout_path(gpmpc_t* g) { return g ? g->h.last_rollout_path : GPMPC_ERR_ARG; }
int gpmpc_last_cluster(gpmpc_t* g) { return g ? g->h.last_cluster : GPMPC_ERR_ARG; }

int gpmpc_last_grad_path(gpmpc_t* g) { return g ? g->h.last_grad_path : GPMPC_ERR_ARG; }

#ifndef GPMPC_BUILD_ID
#define GPMPC_BUILD_ID "unknown"
#endif
const char* gpmpc_build_id(void) { return GPMPC_BUILD_ID; }

int gpmpc_set_cost(gpmpc_t* g, const double* target, const double* W, const double* W_T, double kappa,
                   int clip, const double* smin, const double* smax, int D, int A) {
    if (!g || !target || !W || !W_T) return bad(g, "null argument");
    if (D < 1 || A < 0 || D > kMaxD || D + A > kMaxE) return bad(g, "bad D / A");
    if ((smin == nullptr) != (smax == nullptr)) return bad(g, "state_min and state_max must both be given or both be NULL");
    Handle* h = H_(g);
    GPMPC_HIP_CHECK(h, hipSetDevice(h->device));
    const int DA = D + A;
    const size_t n = (size_t)DA + (size_t)DA * DA + (size_t)D * D + 2 * (size_t)D;
    int rc = grow(h, h->cost, n);
    if (rc) return rc;
    double host[kMaxE + kMaxE * kMaxE + kMaxD * kMaxD + 2 * kMaxD];
    double* q = host;
    memcpy(q, target, DA * sizeof(double)); q += DA;
    memcpy(q, W, (size_t)DA * DA * sizeof(double)); q += DA * DA;
    memcpy(q, W_T, (size_t)D * D * sizeof(double)); q += D * D;
    for (int d = 0; d < D; ++d) q[d] = smin ? smin[d] : 0.0;
    q += D;
    for (int d = 0; d < D; ++d) q[d] = smax ? smax[d] : 0.0;
    GPMPC_HIP_CHECK(h, hipMemcpy(h->cost.p, host, n * sizeof(double), hipMemcpyHostToDevice));
    h->cost_D = D; h->cost_A = A; h->kappa = kappa; h->clip = clip; h->use_constraints = smin ? 1 : 0;
    return GPMPC_OK;
}

static int fill_args(gpmpc_t* g, RolloutArgs& a, const double* actions, const double* mu0, const double* S0,
                     int B, int H, int A, int include_time, double time0, bool need_cost = true) {
    Handle* h = H_(g);
    if (!h->ready) return bad(g, "rollout before prepare / set_factors");
    if (!actions || !mu0 || !S0) return bad(g, "null argument");
    if (B < 1 || H < 1 || A < 0) return bad(g, "need B >= 1, H >= 1");
    if (h->D + A + (include_time ? 1 : 0) != h->E) return bad(g, "D + A (+1 with time) must equal the model's input dim E");
    if (need_cost && (h->cost_D != h->D || h->cost_A != A)) return bad(g, "gpmpc_set_cost not called for this (D, A)");
    memset(&a, 0, sizeof a);
    a.Xt = h->Xt.p; a.beta = h->beta.p; a.Tm = h->Tm.p; a.ils2 = h->ils2.p; a.var = h->var.p; a.logvar = h->logvar.p;
    a.cost = h->cost.p; a.kappa = h->kappa; a.clip = h->clip; a.use_constraints = h->use_constraints;
    a.actions = actions;
    a.N = h->N; a.D = h->D; a.A = A; a.E = h->E; a.H = H; a.B = B;
    a.include_time = include_time; a.time0 = time0;
    memcpy(a.mu0, mu0, h->D * sizeof(double));
    memcpy(a.S0, S0, (size_t)h->D * h->D * sizeof(double));
    return GPMPC_OK;
}

int gpmpc_rollout(gpmpc_t* g, const double* actions, const double* mu0, const double* S0, int B, int H, int A,
                  int include_time, double time0, double* mu_out, double* Sig_out, double* cm_out, double* cv_out,
                  double* J_out, void* stream) {
    Range roctx_range("gpmpc_rollout");
    if (!g) return GPMPC_ERR_ARG;
    RolloutArgs a;
    // the trajectory alone (predict_trajectory, gp_model.py:60-110) needs no cost settings
    int rc = fill_args(g, a, actions, mu0, S0, B, H, A, include_time, time0, cm_out || cv_out || J_out);
    if (rc) return rc;
    GPMPC_HIP_CHECK(H_(g), hipSetDevice(g->h.device));
    a.mu_out = mu_out; a.Sig_out = Sig_out; a.cm_out = cm_out; a.cv_out = cv_out; a.J_out = J_out;
    return launch_rollout(H_(g), a, (hipStream_t)stream);
}

int gpmpc_rollout_grad(gpmpc_t* g, const double* actions, const double* mu0, const double* S0, int B, int H, int A,
                       int include_time, double time0, double* J_out, double* grad_out, double* mu_out, double* Sig_out,
                       double* cm_out, double* cv_out, void* stream) {
    Range roctx_range("gpmpc_rollout_grad");
    if (!g) return GPMPC_ERR_ARG;
    if (!grad_out) return bad(g, "null argument");
    RolloutArgs a;
    int rc = fill_args(g, a, actions, mu0, S0, B, H, A, include_time, time0);
    if (rc) return rc;
    if (A < 1) return bad(g, "gradient needs A >= 1");
    GPMPC_HIP_CHECK(H_(g), hipSetDevice(g->h.device));
    a.mu_out = mu_out; a.Sig_out = Sig_out; a.J_out = J_out; a.cm_out = cm_out; a.cv_out = cv_out;
    return launch_rollout_grad(H_(g), a, grad_out, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------
// One evaluation for a host-side optimiser: the action sequence travels to the device in the forward kernel's argument block (no
// DMA engine start-up, no launch of its own for 200 bytes), the results come back through a pinned, device-mapped host buffer
// written by the last kernel of the evaluation, which then raises a sequence number the host polls.
namespace {
constexpr int kUploadMax = 480;                        // doubles in the argument block (< 4 KiB)
struct UploadArgs { double v[kUploadMax]; };
__global__ __launch_bounds__(64) void upload_kernel(double* dst, int n, const UploadArgs a) {
    for (int i = threadIdx.x; i < n; i += 64) dst[i] = a.v[i];
}
}  // namespace

int gpmpc_objective_grad_host(gpmpc_t* g, const double* actions_host, const double* mu0, const double* S0, int H, int A,
                              int include_time, double time0, const double** result_host, void* stream) {
    Range roctx_range("gpmpc_objective_grad_host");
#if defined(GPMPC_HOST_TIMING)
    static double acc[4] = {0, 0, 0, 0}; static int calls = 0;
    auto now = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; };
    const double T0 = now();
#endif
    if (!g) return GPMPC_ERR_ARG;
    if (!actions_host || !result_host) return bad(g, "null argument");
    Handle* h = H_(g);
    hipStream_t s = (hipStream_t)stream;
    RolloutArgs a;
    int rc = fill_args(g, a, actions_host, mu0, S0, 1, H, A, include_time, time0);
    if (rc) return rc;
    if (A < 1) return bad(g, "gradient needs A >= 1");
    GPMPC_HIP_CHECK(h, hipSetDevice(h->device));
    const int D = h->D, n_act = H * A;
    const size_t n_out = 1 + (size_t)n_act + (size_t)(H + 1) * (D + D * D + 2);
    rc = grow(h, h->hio, (size_t)n_act + n_out);
    if (rc) return rc;
    if (h->hio_host_cap < n_out) {
        if (h->hio_host) GPMPC_HIP_CHECK(h, hipHostFree(h->hio_host));
        h->hio_host = nullptr; h->hio_host_dev = nullptr; h->hio_host_cap = 0;
        GPMPC_HIP_CHECK(h, hipHostMalloc(reinterpret_cast<void**>(&h->hio_host), n_out * sizeof(double), hipHostMallocMapped));
        GPMPC_HIP_CHECK(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&h->hio_host_dev), h->hio_host, 0));
        h->hio_host_cap = n_out;
    }
    if (!h->hio_flag) {
        GPMPC_HIP_CHECK(h, hipHostMalloc(reinterpret_cast<void**>(&h->hio_flag), 64, hipHostMallocMapped));
        GPMPC_HIP_CHECK(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&h->hio_flag_dev), h->hio_flag, 0));
        *h->hio_flag = 0;
    }
    double* act_dev = h->hio.p;
    double* out_dev = act_dev + n_act;
    a.act_inline_n = 0;
    if (n_act <= kInlineActs) {                 // in the forward kernel's argument block (launch_rollout uploads for the other paths)
        a.act_inline_n = n_act;
        a.act_store = act_dev;
        memcpy(a.act_inline, actions_host, n_act * sizeof(double));
    } else if (n_act <= kUploadMax) {
        UploadArgs u;
        memcpy(u.v, actions_host, n_act * sizeof(double));
        hipLaunchKernelGGL(upload_kernel, dim3(1), dim3(64), 0, s, act_dev, n_act, u);
        GPMPC_HIP_CHECK(h, hipGetLastError());
    } else {
        GPMPC_HIP_CHECK(h, hipMemcpyAsync(act_dev, actions_host, n_act * sizeof(double), hipMemcpyHostToDevice, s));
    }
    a.actions = act_dev;
    double* q = out_dev;
    a.J_out = q; q += 1;
    double* grad = q; q += n_act;
    a.mu_out = q; q += (size_t)(H + 1) * D;
    a.Sig_out = q; q += (size_t)(H + 1) * D * D;
    a.cm_out = q; q += H + 1;
    a.cv_out = q;
    // The reverse sweep -- the last kernel of the evaluation -- copies the results to the pinned mirror and then raises this call's
    // sequence number there; the host polls that word (a stream synchronisation costs ~10 us more than the store takes to arrive)
    // and asks the stream now and then, so that a failed launch ends the wait.
    const unsigned long long seq = ++h->hio_seq;
    const bool sweep_exports = D <= 8;               // (the wide-state sweep, 8 < D <= 16, has no export: a synchronisation and a copy)
    h->hx_out = h->hio_host_dev; h->hx_src = out_dev; h->hx_n = sweep_exports ? (int)n_out : 0;
#if defined(GPMPC_HOST_TIMING)
    const double T1 = now();
    g_host_timing_fwd = 0.0;
#endif
    rc = launch_rollout_grad(h, a, grad, s);
    h->hx_n = 0;
    if (rc) return rc;
    if (!sweep_exports) {
        GPMPC_HIP_CHECK(h, hipStreamSynchronize(s));
        GPMPC_HIP_CHECK(h, hipMemcpy(h->hio_host, out_dev, n_out * sizeof(double), hipMemcpyDeviceToHost));
        *result_host = h->hio_host;
        return GPMPC_OK;
    }
#if defined(GPMPC_HOST_TIMING)
    const double T2 = now();
#endif
    volatile unsigned long long* flag = h->hio_flag;
    for (unsigned spins = 1; *flag != seq; ++spins) {
        __builtin_ia32_pause();
        if ((spins & 0xfff) == 0) {
            const hipError_t qe = hipStreamQuery(s);
            if (qe == hipSuccess) {
                if (*flag == seq) break;
                h->err = "objective_grad_host: the launches ended without the sweep's completion word";
                return GPMPC_ERR_HIP;
            }
            if (qe != hipErrorNotReady) GPMPC_HIP_CHECK(h, qe);
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
#if defined(GPMPC_HOST_TIMING)
    const double T3 = now();
    acc[0] += T1 - T0; acc[1] += g_host_timing_fwd - T1; acc[2] += T2 - g_host_timing_fwd; acc[3] += T3 - T2;
    if (++calls % 200 == 0) {
        fprintf(stderr, "HOST TIMING us: set-up %.2f | plan + forward launch %.2f | remaining launches %.2f | wait %.2f\n", acc[0] / 200, acc[1] / 200, acc[2] / 200, acc[3] / 200);
        acc[0] = acc[1] = acc[2] = acc[3] = 0;
    }
#endif
    *result_host = h->hio_host;
    return GPMPC_OK;
}

int gpmpc_cem_search(gpmpc_t* g, const double* mu0, const double* S0, int B, int H, int A, int include_time, double time0,
                     int iterations, int n_elite, unsigned long long seed, const double* first_candidate, int mapper,
                     const double* max_change, const double* action_prev, const double* noise_dev, double* best_out_dev,
                     void* stream) {
    Range roctx_range("gpmpc_cem_search");
    if (!g) return GPMPC_ERR_ARG;
    if (!best_out_dev) return bad(g, "null argument");
    Handle* h = H_(g);
    if (!h->ready) return bad(g, "search before prepare / set_factors");
    // fill_args wants an actions pointer; the search supplies its own device buffer afterwards
    RolloutArgs a;
    int rc = fill_args(g, a, best_out_dev, mu0, S0, B, H, A, include_time, time0);
    if (rc) return rc;
    GPMPC_HIP_CHECK(h, hipSetDevice(h->device));
    return run_cem_search(h, a, iterations, n_elite, seed, first_candidate, mapper, max_change, action_prev, noise_dev,
                          best_out_dev, (hipStream_t)stream);
}

int gpmpc_cem_local(gpmpc_t* g, const double* mu0, const double* S0, int B_total, int first, int B_local, int H, int A,
                    int include_time, double time0, int iteration, int n_elite, unsigned long long seed,
                    const double* first_candidate, int mapper, const double* max_change, const double* action_prev,
                    const double* noise_dev, const double* state_dev, double* elites_out_dev, void* stream) {
    Range roctx_range("gpmpc_cem_local");
    if (!g) return GPMPC_ERR_ARG;
    if (!state_dev || !elites_out_dev) return bad(g, "null argument");
    Handle* h = H_(g);
    if (!h->ready) return bad(g, "search before prepare / set_factors");
    if (B_local < 0) return bad(g, "negative slice length");
    RolloutArgs a;
    int rc = fill_args(g, a, state_dev, mu0, S0, B_local > 0 ? B_local : 1, H, A, include_time, time0);
    if (rc) return rc;
    a.B = B_local;
    GPMPC_HIP_CHECK(h, hipSetDevice(h->device));
    return run_cem_local(h, a, B_total, first, iteration, n_elite, seed, first_candidate, mapper, max_change, action_prev,
                         noise_dev, state_dev, elites_out_dev, (hipStream_t)stream);
}

int gpmpc_cem_merge(gpmpc_t* g, const double* elites_dev, int lists, int n_elite, int n, int iteration, double* state_dev,
                    void* stream) {
    Range roctx_range("gpmpc_cem_merge");
    if (!g) return GPMPC_ERR_ARG;
    if (!elites_dev || !state_dev) return bad(g, "null argument");
    Handle* h = H_(g);
    GPMPC_HIP_CHECK(h, hipSetDevice(h->device));
    return run_cem_merge(h, elites_dev, lists, n_elite, n, iteration, state_dev, (hipStream_t)stream);
}

int gpmpc_rollout_timed(gpmpc_t* g, const double* actions, const double* mu0, const double* S0, int B, int H, int A,
                        int include_time, double time0, double* J_out, int reps, float* ms, void* stream) {
    if (!g || !ms || reps < 1) return GPMPC_ERR_ARG;
    Handle* h = H_(g);
    RolloutArgs a;
    int rc = fill_args(g, a, actions, mu0, S0, B, H, A, include_time, time0);
    if (rc) return rc;
    GPMPC_HIP_CHECK(h, hipSetDevice(h->device));
    a.J_out = J_out;
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    GPMPC_HIP_CHECK(h, hipEventCreate(&e0));
    GPMPC_HIP_CHECK(h, hipEventCreate(&e1));
    GPMPC_HIP_CHECK(h, hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) {
        rc = launch_rollout(h, a, s);
        if (rc) break;
    }
    GPMPC_HIP_CHECK(h, hipEventRecord(e1, s));
    GPMPC_HIP_CHECK(h, hipEventSynchronize(e1));
    float t = 0.f;
    GPMPC_HIP_CHECK(h, hipEventElapsedTime(&t, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms = t / (float)reps;
    return rc;
}

int gpmpc_argmin(gpmpc_t* g, const double* J, int B, long long first, double* best_J, long long* best_idx, void* stream) {
    Range roctx_range("gpmpc_argmin");
    if (!g || !J || B < 1 || first < 0) return bad(g, "bad argument");
    Handle* h = H_(g);
    GPMPC_HIP_CHECK(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    int rc = launch_argmin(h, J, B, first, s);
    if (rc) return rc;
    double out[2];
    GPMPC_HIP_CHECK(h, hipMemcpyAsync(out, h->best.p, sizeof out, hipMemcpyDeviceToHost, s));
    GPMPC_HIP_CHECK(h, hipStreamSynchronize(s));
    if (best_J) *best_J = out[0];
    if (best_idx) memcpy(best_idx, &out[1], sizeof(long long));
    return GPMPC_OK;
}

int gpmpc_argmin_async(gpmpc_t* g, const double* J, int B, long long first, const double* actions, int HA, double* out_dev,
                       void* stream) {
    Range roctx_range("gpmpc_argmin_async");
    if (!g || !J || !out_dev || B < 1 || first < 0 || HA < 0) return bad(g, "bad argument");
    Handle* h = H_(g);
    GPMPC_HIP_CHECK(h, hipSetDevice(h->device));
    return launch_argmin_to(h, J, B, first, actions, HA, out_dev, (hipStream_t)stream);
}

}  // extern "C"
