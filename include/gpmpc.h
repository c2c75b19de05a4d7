/*
 * gpmpc.h -- C ABI of the MI355X-native GP-MPC hot path (libgpmpc_hip.so).
 *
 * The reference (SimonRennotte/Data-Efficient-RL-with-Probabilistic-MPC) has no FFI:
 * its boundary is the duck-typed Python interface AbstractStateTransitionModel
 * (rl_gp_mpc/control_objects/models/abstract_model.py:5-28) plus the candidate loop
 * of GpMpcController._get_optimal_actions (controllers/gp_mpc_controller.py:114-153).
 * Each entry point below names the reference code it replaces.  The Python mirror
 * of those classes (package `..._amd/control_objects/`) calls ONLY these functions
 * for arithmetic; INTEGRATION.md shows the ctypes stub a reference maintainer adds.
 *
 * Conventions
 *   - every array is fp64 (the reference forces fp64: config_classes/total_config.py:11),
 *     row-major, contiguous;
 *   - `*_dev` pointers are DEVICE pointers on the handle's GPU (e.g. tensor.data_ptr()),
 *     `*_host` pointers are HOST pointers read before the call returns;
 *   - `stream` is a hipStream_t (NULL = default stream).  Calls are asynchronous on that
 *     stream unless stated otherwise; the caller owns all in/out buffers, the handle owns
 *     its workspace (factor matrices, per-candidate scratch);
 *   - return value: GPMPC_OK or a negative error; gpmpc_last_error() gives the text;
 *   - one handle per device; a handle is not thread-safe; handles are independent.
 */
#ifndef GPMPC_H
#define GPMPC_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gpmpc gpmpc_t;

enum {
    GPMPC_OK = 0,
    GPMPC_ERR_ARG = -1,    /* bad argument / shape / state                                   */
    GPMPC_ERR_NOT_PD = -2, /* Cholesky hit a non-positive pivot (reference: uncaught          */
                           /* torch.linalg.cholesky error, models/gp_model.py:427)            */
    GPMPC_ERR_HIP = -3,    /* HIP runtime error                                               */
    GPMPC_ERR_LIMIT = -4   /* shape outside compiled limits (D <= 16, D+A+time <= 24)          */
};

#define GPMPC_MAX_D 16
#define GPMPC_MAX_E 24

/* Library ABI version (bumped on any signature change). */
int gpmpc_abi_version(void);

/* Identifier of the build: a hash over the library's sources (csrc/Makefile), so that counter files kept under
 * profiles/ can name the build they were collected on (bench.py nulls counter-derived figures of another build). */
const char* gpmpc_build_id(void);

/* Create / destroy a handle bound to HIP device `device_id`. */
int gpmpc_create(gpmpc_t** out, int device_id);
int gpmpc_destroy(gpmpc_t* h);
const char* gpmpc_last_error(const gpmpc_t* h);

/*
 * gpmpc_prepare  <->  GpStateTransitionModel.prepare_inference + calculate_factorizations
 *                     (models/gp_model.py:182-191, 400-431).
 * Builds, per output dimension a: K_a = outputscale_a * exp(-1/2 sum_e ((x_e-x'_e)/l_ae)^2)
 * (gpytorch ScaleKernel(RBFKernel(ard)), gp_model.py:391,425), L_a = chol(K_a + noise_a I)
 * (:427), iK_a = (L_a L_a^T)^-1 (:428), beta_a = iK_a y_a (:429-430), and the derived
 * tables the rollout kernel streams.  Synchronises `stream` before returning so that a
 * failed factorisation is reported here (GPMPC_ERR_NOT_PD; failing GP via
 * gpmpc_last_error()).
 *   X_dev (N,E)  Y_dev (N,D)  lengthscales_dev (D,E)  outputscales_dev (D)  noises_dev (D)
 */
int gpmpc_prepare(gpmpc_t* h, const double* X_dev, const double* Y_dev,
                  const double* lengthscales_dev, const double* outputscales_dev,
                  const double* noises_dev, int N, int D, int E, void* stream);

/*
 * Hyper-parameter training objective: for each of the D GPs the negative exact marginal
 * log-likelihood per data point, -log p(y_a | X, theta_a) / N, and its gradient wrt the lengthscales, the
 * outputscale and the noise variance -- the loss gpytorch's ExactMarginalLogLikelihood and autograd give the
 * reference's LBFGS loop (rl_gp_mpc/control_objects/models/gp_model.py:262-275).  Arguments as gpmpc_prepare;
 * out_host (D, E + 3) = [loss | d/d lengthscale_e (E) | d/d outputscale | d/d noise], host memory, synchronous.
 * Always factorises from scratch and REPLACES the handle's cached factors with those of these hyper-parameters:
 * give the training loop a handle of its own.  GPMPC_ERR_NOT_PD as gpmpc_prepare.
 */
int gpmpc_mll(gpmpc_t* h, const double* X_dev, const double* Y_dev, const double* lengthscales_dev,
              const double* outputscales_dev, const double* noises_dev, int N, int D, int E, double* out_host, void* stream);

/* How the last gpmpc_prepare obtained its factors: 0 = full factorisation, 1 = border update of the cached
 * factors (the new memory was the cached one plus <= 8 appended points, hyper-parameters unchanged; O(k N^2)),
 * 2 = cache hit (nothing changed).  The reference refactorises at every control step
 * (gp_mpc_controller.py:117).  Option "incremental" (default 1) switches the reuse off, "refresh_every"
 * (default 32) bounds the number of border updates between full factorisations. */
int gpmpc_last_prepare_mode(gpmpc_t* h);

/* Which kernels the last gpmpc_rollout / _grad / _cem_search launch used for the forward pass: 0 = the fused-horizon
 * kernel (one workgroup per candidate, all H steps in one launch), 1 = the streaming kernel (per-point arrays beyond
 * the LDS), 2 = the batch-major path (per horizon step: the N x N work of the diagonal output pairs, gp_model.py:161-175,
 * by workgroups that own a 128 x 128 tile of beta beta^T - iK and loop over the candidates, + a per-candidate step
 * kernel) -- chosen when D <= 4, the D tables exceed an XCD's L2 and B >= 512; option "pair_tiles" 1 forces, 2 forbids it,
 * "tile_chunk" sets the candidates per tile workgroup. */
int gpmpc_last_rollout_path(gpmpc_t* h);

/* Workgroups per candidate of the last fused-horizon launch.  1 = one workgroup per candidate; > 1 = the few-candidate
 * cooperative form: the reference evaluates ONE action sequence per objective call (restarts_optim 1-2,
 * gp_mpc_controller.py:125-141), so while candidates x cluster fit the chip a cluster of workgroups shares each candidate's
 * horizon step (bit-identical trajectories).  Option "cluster": 0 auto, 1 never, 2..32 fixed. */
int gpmpc_last_cluster(gpmpc_t* h);

/* Which moment passes the last gpmpc_rollout_grad launched (bit mask): 1 = off-diagonal output pairs in separable form on
 * the matrix cores, 2 = diagonal pairs batch-major over all (candidate, step) items, 4 = the streaming element-wise pass
 * (per-point arrays beyond the LDS), 8 = the 8 < D <= 16 pass, 16 (with 2) = the diagonal pairs' moments were formed by the
 * batch-major FORWARD's own tile pass (one evaluation of the pairwise weights serves the forward sums and the moments; option
 * "grad_fuse" 0 switches that off), 32 = the mean part of the streaming pass by its own kernel (lanes over points; option
 * "grad_mean" 0: inside the pass).  0 = the element-wise pass alone.  A test hook: lets a parity
 * test assert that it measured the dispatch that ships (gp_mpc_controller.py:277 is one autograd call in the reference). */
int gpmpc_last_grad_path(gpmpc_t* h);

/*
 * Same cached state as gpmpc_prepare but with iK (D,N,N) and beta (D,N) supplied by the
 * caller (test hook: lets the rollout kernel be checked in isolation from the
 * factorisation; also the entry for callers that keep their own factorisation).
 */
int gpmpc_set_factors(gpmpc_t* h, const double* X_dev, const double* iK_dev, const double* beta_dev,
                      const double* lengthscales_dev, const double* outputscales_dev,
                      int N, int D, int E, void* stream);

/* Borrowed device pointers to iK (D,N,N) and beta (D,N); valid until the next
 * prepare/set_factors/destroy.  <-> attributes self.iK, self.beta (gp_model.py:187). */
int gpmpc_get_factors(gpmpc_t* h, const double** iK_dev, const double** beta_dev);

/* Copies of the cached factors into caller-owned device buffers (either may be NULL):
 * iK_dst_dev (D,N,N), beta_dst_dev (D,N).  Asynchronous on `stream`. */
int gpmpc_read_factors(gpmpc_t* h, double* iK_dst_dev, double* beta_dst_dev, void* stream);

/* Options (measurement / test hooks): "threads" (rollout workgroup size: 0 = auto, 256/512/1024),
 * "rows_per_chunk", "force_path" (0 auto / 1 direct exp / 2 element-wise Taylor / 4 tabulated mid-range exp in the matrix-core pair pass), "force_separable",
 * "force_global_scratch" (0/1, the large-N streaming kernel at any N), "cols_per_lane" (0 auto, 1, 2: columns per
 * lane in the pairwise pass of the rollout kernel), "grad_cols_per_lane" (same for the gradient's moment pass),
 * "exact_dim" (2: never the compile-time-D kernel instantiation),
 * "incremental" (0/1, default 1), "refresh_every" (default 32), "fused_prepare" (0/1, default 1: memories of
 * up to 240 points (the measured crossover) are factorised by one launch, one workgroup per GP; 0 = the panel-by-panel path),
 * and, for memories of "outer_min_n" (default 640, the measured crossover) points and more (defaults = the shipped path,
 * the others are A/B and test hooks):
 * "outer_block" (1; 0: 32-wide panels only), "tile128" (1: 128 x 128 tiles with 8 wavefronts for the tiled
 * products; 0: 64 x 64), "outer2" (2: binary outer levels of the trailing update up to 128 * 2^value columns),
 * "block128" (1: a whole 128-column outer panel in two launches; 0: 32-column panels), "inner_left" (1: those
 * panels left-looking inside the outer panel; 0: right-looking).
 * Round 3: "pair_tiles" (0 auto / 1 always / 2 never: the batch-major rollout path -- one set of launches per horizon
 * step, workgroups own 128 x 128 tiles of T_a and loop over candidates; auto: D <= 4, 4 D N^2 >= 6e6, B >= 2 x CUs),
 * "tile_chunk" (candidates per tile workgroup, 0 = chosen to fill the last round), "tile_overlap" (1: point pass on a side
 * stream), "grad_separable" and "grad_tiles" (0 never / 1 auto / 2 always: see gpmpc_rollout_grad).
 * Round 4: "lds_limit_kb" (LDS budget of a fused-horizon workgroup; with "threads" 512 two workgroups share a CU -- chosen
 * automatically for 64 < N <= 256 from 8 workgroups per CU on), "prepare_overlap" (1 / 0: inverse chain of the 32-wide panel path
 * on a side stream), "grad_mean" (1 / 0: mean part of the moment pass by its own kernel, D <= 4), "grad_fuse" (1 / 0: gradient launches whose forward takes the batch-major path form the diagonal pairs' tile moments
 * inside the forward's tile pass), "grad_chunk_rows" (rows per work item of the LDS-resident moment pass: 0 = chosen by the host's
 * schedule model, csrc/moment_schedule.h; else a multiple of 4 up to 64 -- GPMPC_ERR_ARG otherwise), "grad_share_cu" (that pass
 * at D <= 3 as two 512-thread workgroups per CU: 0 auto / 1 where the LDS fits twice / 2 never). */
int gpmpc_set_option(gpmpc_t* h, const char* name, long long value);

/*
 * Quadratic cost of SetpointStateRewardMapper
 * (states_reward_mappers/setpoint_distance_reward_mapper.py:12-68,124-142) and the LCB
 * objective settings of compute_mean_lcb_trajectory (gp_mpc_controller.py:270-276).
 *   target_host (D+A)   W_host (D+A,D+A)   W_T_host (D,D)   kappa = exploration_factor
 *   clip_to_zero  = reward.clip_lower_bound_cost_to_0
 *   state_min_host/state_max_host (D) or NULL = reward.use_constraints False
 */
int gpmpc_set_cost(gpmpc_t* h, const double* target_host, const double* W_host,
                   const double* W_T_host, double kappa, int clip_to_zero,
                   const double* state_min_host, const double* state_max_host, int D, int A);

/*
 * gpmpc_rollout  <->  B x [ predict_trajectory (gp_model.py:60-110, H calls of
 * predict_next_state_change :112-180) + get_rewards_trajectory
 * (setpoint_distance_reward_mapper.py:144-149) + forward value of
 * compute_mean_lcb_trajectory (gp_mpc_controller.py:267-276) ], one candidate action
 * sequence per workgroup, all H steps inside one launch.
 *   actions_dev (B,H,A) model-space actions in [0,1]
 *   mu0_host (D), S0_host (D,D): initial state distribution (same for all candidates)
 *   include_time / time0: ModelConfig.include_time_model, current_time_idx (gp_model.py:101-102)
 * Outputs (each may be NULL to skip the store):
 *   mu_out_dev (B,H+1,D)  Sig_out_dev (B,H+1,D,D)   index 0 = input state (gp_model.py:91-92)
 *   cost_mu_out_dev (B,H+1) = -rewards   cost_var_out_dev (B,H+1)
 *   J_out_dev (B) = mean-LCB objective the optimiser / argmin sees
 * The three cost outputs need gpmpc_set_cost for this (D, A); with all three NULL the call is the
 * plain predict_trajectory and needs no cost settings.
 */
int gpmpc_rollout(gpmpc_t* h, const double* actions_dev, const double* mu0_host,
                  const double* S0_host, int B, int H, int A, int include_time, double time0,
                  double* mu_out_dev, double* Sig_out_dev, double* cost_mu_out_dev,
                  double* cost_var_out_dev, double* J_out_dev, void* stream);

/*
 * gpmpc_argmin  <->  the keep-the-best rule of gp_mpc_controller.py:146-148 applied to a
 * vector of objective values: first strict minimum wins; a NaN in GLOBAL slot 0 is adopted
 * and never displaced; any other NaN is never selected.  `first_global_index` is the global
 * index of J_dev[0] (0 on one GPU; the shard offset when candidates are sharded over GPUs),
 * the returned index is global.  A shard with no selectable value returns index -1, J = +inf.
 * Synchronises `stream`; results are written to HOST memory.
 */
int gpmpc_argmin(gpmpc_t* h, const double* J_dev, int B, long long first_global_index,
                 double* best_J_host, long long* best_idx_host, void* stream);

/*
 * Objective AND its analytic gradient for B candidates:  J_out_dev (B),  grad_out_dev (B, H, A) = dJ/d(actions),
 * the quantity the reference obtains with `mean_cost.backward()` (gp_mpc_controller.py:277) and hands to
 * scipy as `jac` (:132-139, :285).  Three launches on `stream`: the forward rollout (as gpmpc_rollout),
 * the pairwise moment pass (one workgroup per candidate and horizon step) and the reverse sweep.
 * clip_lower_bound_cost_to_0 clips the value only (the reference's clamp passes the gradient through).
 * mu_out_dev / Sig_out_dev / cost_mu_out_dev / cost_var_out_dev as in gpmpc_rollout (nullable).  Supported for D <= 8 with A (+1 with
 * time) <= 6 and for 8 < D <= 16 (matrix-core moment pass + pair-walking sweep: config 5); memories whose per-point arrays
 * exceed the LDS take a streaming moment pass (N up to ~15 000 at D = 4, ~2 000 at D = 8: the column-factor array and the
 * point chunks must fit); otherwise GPMPC_ERR_LIMIT (callers then difference gpmpc_rollout).  For D <= 4 the moment pass is
 * split by pair kind where that pays: off-diagonal pairs in separable form on the matrix cores (option "grad_separable"),
 * diagonal pairs batch-major over all (candidate, step) items at large N x large B (option "grad_tiles"); the results do
 * not depend on the split beyond rounding (1e-7 of the gradient's scale against the reference's autograd in every form).
 */
int gpmpc_rollout_grad(gpmpc_t* h, const double* actions_dev, const double* mu0_host, const double* S0_host,
                       int B, int H, int A, int include_time, double time0, double* J_out_dev, double* grad_out_dev,
                       double* mu_out_dev, double* Sig_out_dev, double* cost_mu_out_dev, double* cost_var_out_dev,
                       void* stream);

/*
 * gpmpc_objective_grad_host  <->  ONE call of compute_mean_lcb_trajectory (gp_mpc_controller.py:229-285) as scipy's L-BFGS-B
 * makes it (:133-141: one action sequence per evaluation, host arrays in, (float, host gradient) out): gpmpc_rollout_grad for
 * B = 1 with host buffers on both sides and ONE synchronisation of `stream`.
 *   actions_host (H, A) model-space actions
 *   *result_host: pinned host buffer owned by the handle, valid until the next call:
 *       J (1) | grad (H, A) | mu (H+1, D) | Sig (H+1, D, D) | cost_mu (H+1) | cost_var (H+1)
 * (the trajectory and the stage costs are what the reference caches on the controller for IterationInformation, :279-283).
 */
int gpmpc_objective_grad_host(gpmpc_t* h, const double* actions_host, const double* mu0_host, const double* S0_host,
                              int H, int A, int include_time, double time0, const double** result_host, void* stream);

/* Asynchronous form of gpmpc_argmin for the multi-GPU path: same rule, no host synchronisation; writes
 * the record out_dev[0] = best J, out_dev[1] = (double) global index (-1.0 if nothing selectable) and, when
 * actions_dev (B, HA) is given, out_dev[2 .. 2+HA) = the winning action sequence, on `stream` -- ready to
 * be all-gathered over RCCL as is. */
int gpmpc_argmin_async(gpmpc_t* h, const double* J_dev, int B, long long first_global_index,
                       const double* actions_dev, int HA, double* out_dev, void* stream);

/*
 * Candidate optimisation whose loop stays on the device (replaces B sequential scipy restarts,
 * gp_mpc_controller.py:125-141, by a cross-entropy search over the same box [0,1]^(H*A)): per iteration
 * B optimiser vectors are drawn around the current mean / std (iteration 0: uniform; slot 0 = the incumbent, in
 * iteration 0 `first_candidate_host` when given), mapped to model actions (mapper 0: identity reshape,
 * normalization_action_mapper.py:21-23; 1: scaled deltas + cumulative sum + pass-through clamp,
 * derivative_action_mapper.py:28-35, with max_change_host (A) and action_prev_host (A)), evaluated by one rollout
 * launch, and the n_elite best refit mean and std.  All launches are enqueued on `stream`; NOTHING is read back
 * between iterations.  best_out_dev (H*A + 1) = [best optimiser vector | its objective], valid after `stream`
 * has been synchronised.  Draws come from Philox4x32-10 keyed by `seed`, or from noise_dev (iterations, B, H*A)
 * when given (iteration 0: uniforms in [0,1); later: standard normals) -- the hook the parity test uses.
 * 2 <= B <= 4096.
 */
int gpmpc_cem_search(gpmpc_t* h, const double* mu0_host, const double* S0_host, int B, int H, int A,
                     int include_time, double time0, int iterations, int n_elite, unsigned long long seed,
                     const double* first_candidate_host, int mapper, const double* max_change_host,
                     const double* action_prev_host, const double* noise_dev, double* best_out_dev, void* stream);

/*
 * The same search with the candidates SHARDED over GPUs (the reference's restart loop gp_mpc_controller.py:125-141
 * is what is being spread): one iteration in two halves, with ONE exchange between them.
 *   gpmpc_cem_local: this GPU's slice [first, first + B_local) of the B_total candidates of iteration `iteration` -- draws (the
 *     Philox counters / the rows of noise_dev (iterations, B_total, H*A) are indexed by the GLOBAL candidate, so the union of
 *     the slices IS the population gpmpc_cem_search draws), mapper, one rollout launch -- and the slice's n_elite best as records
 *     elites_out_dev (n_elite, 2 + H*A) = [J (NaN -> +inf) | global index | optimiser vector], sorted; shorter slices pad with
 *     (+inf, INT_MAX) records.  B_local = 0 is allowed (more GPUs than candidates).
 *   (the caller all-gathers the records of all GPUs: RCCL, n_elite (2 + H*A) doubles per GPU)
 *   gpmpc_cem_merge: elites_dev (lists * n_elite, 2 + H*A) -> state_dev = [mean (n) | std (n) | best vector (n) | best J], n = H*A:
 *     the sort, incumbent rule and elite statistics of gpmpc_cem_search on the union, in the same summation order -- every GPU
 *     holds the same state afterwards, bit for bit the single-GPU search's.  lists * n_elite <= 4096.
 * state_dev is read by gpmpc_cem_local from iteration 1 on (iteration 0 draws uniformly and ignores it).  Nothing synchronises.
 */
int gpmpc_cem_local(gpmpc_t* h, const double* mu0_host, const double* S0_host, int B_total, int first, int B_local, int H, int A,
                    int include_time, double time0, int iteration, int n_elite, unsigned long long seed,
                    const double* first_candidate_host, int mapper, const double* max_change_host,
                    const double* action_prev_host, const double* noise_dev, const double* state_dev,
                    double* elites_out_dev, void* stream);
int gpmpc_cem_merge(gpmpc_t* h, const double* elites_dev, int lists, int n_elite, int n, int iteration, double* state_dev,
                    void* stream);

/* Kernel-only timing helper for bench.py: runs `reps` rollouts back to back on `stream`
 * bracketed by HIP events recorded on THAT stream and returns the average milliseconds
 * per launch in *ms_host (outputs as gpmpc_rollout; synchronises). */
int gpmpc_rollout_timed(gpmpc_t* h, const double* actions_dev, const double* mu0_host,
                        const double* S0_host, int B, int H, int A, int include_time, double time0,
                        double* J_out_dev, int reps, float* ms_host, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GPMPC_H */
