/*
 * gpmpc.h -- C ABI of the MI355X-native GP-MPC hot path (libgpmpc_hip.so).
 *
 * The reference (SimonRennotte/Data-Efficient-RL-with-Probabilistic-MPC) has no FFI: its boundary is the duck-typed Python
 * interface AbstractStateTransitionModel (rl_gp_mpc/control_objects/models/abstract_model.py:5-28) plus the candidate loop of
 * GpMpcController._get_optimal_actions (controllers/gp_mpc_controller.py:114-153).  Each entry point names the reference code it
 * replaces.  The Python mirror of those classes (package `..._amd/control_objects/`) calls ONLY these functions for arithmetic;
 * INTEGRATION.md shows the ctypes stub a reference maintainer adds.
 *
 * Conventions: every array is fp64 (the reference forces fp64: config_classes/total_config.py:11), row-major, contiguous;
 * `*_dev` = DEVICE pointers on the handle's GPU, `*_host` = HOST pointers read before the call returns; `stream` = hipStream_t
 * (NULL = default stream), calls are asynchronous on it unless stated; the caller owns all in / out buffers, the handle its
 * workspace; return GPMPC_OK or a negative error, gpmpc_last_error() gives the text; one handle per device, not thread-safe.
 */
#ifndef GPMPC_H
#define GPMPC_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct gpmpc gpmpc_t;

enum {
    GPMPC_OK = 0,
    GPMPC_ERR_ARG = -1,    /* bad argument / shape / state */
    GPMPC_ERR_NOT_PD = -2, /* Cholesky hit a non-positive pivot (reference: uncaught torch.linalg.cholesky error, gp_model.py:427) */
    GPMPC_ERR_HIP = -3,    /* HIP runtime error */
    GPMPC_ERR_LIMIT = -4   /* shape outside compiled limits (D <= 16, D + A + time <= 24) */
};
#define GPMPC_MAX_D 16
#define GPMPC_MAX_E 24

int gpmpc_abi_version(void);            /* bumped on any signature change */
const char* gpmpc_build_id(void);       /* hash over the library's sources: counter files under profiles/ name their build */
int gpmpc_create(gpmpc_t** out, int device_id);
int gpmpc_destroy(gpmpc_t* h);
const char* gpmpc_last_error(const gpmpc_t* h);

/*
 * gpmpc_prepare  <->  GpStateTransitionModel.prepare_inference + calculate_factorizations (gp_model.py:182-191, 400-431): per
 * output a, K_a = outputscale_a exp(-1/2 sum_e ((x_e - x'_e) / l_ae)^2) (gpytorch ScaleKernel(RBFKernel(ard)), :391, :425),
 * L_a = chol(K_a + noise_a I) (:427), iK_a (:428), beta_a = iK_a y_a (:429-430) and the tables the rollout streams.
 *   X_dev (N,E)  Y_dev (N,D)  lengthscales_dev (D,E)  outputscales_dev (D)  noises_dev (D)
 * Synchronises `stream` (a failed factorisation is reported here: GPMPC_ERR_NOT_PD).  Reuse across control steps (the reference
 * refactorises every step, gp_mpc_controller.py:117): gpmpc_last_prepare_mode = 0 full factorisation, 1 border update (the
 * cached memory plus <= 8 appended points, same hyper-parameters), 2 cache hit.
 */
int gpmpc_prepare(gpmpc_t* h, const double* X_dev, const double* Y_dev, const double* lengthscales_dev,
                  const double* outputscales_dev, const double* noises_dev, int N, int D, int E, void* stream);
int gpmpc_last_prepare_mode(gpmpc_t* h);

/* Training objective: per GP -log p(y_a | X, theta_a) / N and its gradient wrt lengthscales, outputscale and
 * noise -- what gpytorch's ExactMarginalLogLikelihood + autograd give the reference's LBFGS loop (gp_model.py:262-275).
 * out_host (D, E + 3) = [loss | d/d lengthscale (E) | d/d outputscale | d/d noise]; synchronous; REPLACES the cached factors. */
int gpmpc_mll(gpmpc_t* h, const double* X_dev, const double* Y_dev, const double* lengthscales_dev,
              const double* outputscales_dev, const double* noises_dev, int N, int D, int E, double* out_host, void* stream);

/* The cached state with iK (D,N,N), beta (D,N) supplied by the caller (test hook; callers that keep their own factorisation);
 * borrowed pointers to / copies of the cached factors  <->  attributes self.iK, self.beta (gp_model.py:187). */
int gpmpc_set_factors(gpmpc_t* h, const double* X_dev, const double* iK_dev, const double* beta_dev,
                      const double* lengthscales_dev, const double* outputscales_dev, int N, int D, int E, void* stream);
int gpmpc_get_factors(gpmpc_t* h, const double** iK_dev, const double** beta_dev);
int gpmpc_read_factors(gpmpc_t* h, double* iK_dst_dev, double* beta_dst_dev, void* stream);

/*
 * Options.  Behaviour: "incremental" (0/1, default 1: reuse / border-update the cached factors), "refresh_every" (32: border
 * updates between full factorisations), "cluster" (few-candidate cooperative form: 0 auto, 1 never, 2..32 workgroups per
 * candidate), "threads" (fused-horizon workgroup: 0 auto, 256 / 512 / 1024), "pair_tiles" (batch-major rollout path: 0 auto,
 * 1 always, 2 never).  Dispatch hooks of the parity tests: "rows_per_chunk", "cols_per_lane", "force_path" (1 direct exp,
 * 2 element-wise Taylor), "force_separable", "force_global_scratch", "grad_separable" / "grad_tiles" / "grad_stream" /
 * "grad_mean" / "grad_share_cu" / "grad_chunk_rows", "fused_prepare", "outer_min_n".  Measurement (A/B) switches of single
 * kernels are listed with their measurements in csrc/gpmpc_internal.h (struct Handle, opt_*).  Unknown names: GPMPC_ERR_ARG.
 */
int gpmpc_set_option(gpmpc_t* h, const char* name, long long value);

/* Quadratic cost of SetpointStateRewardMapper (setpoint_distance_reward_mapper.py:12-68, 124-142) and the LCB settings of
 * compute_mean_lcb_trajectory (gp_mpc_controller.py:270-276): target_host (D+A), W_host (D+A,D+A), W_T_host (D,D),
 * kappa = exploration_factor, clip_to_zero = clip_lower_bound_cost_to_0, state_min/max_host (D) or NULL = use_constraints False. */
int gpmpc_set_cost(gpmpc_t* h, const double* target_host, const double* W_host, const double* W_T_host, double kappa,
                   int clip_to_zero, const double* state_min_host, const double* state_max_host, int D, int A);

/*
 * gpmpc_rollout  <->  B x [ predict_trajectory (gp_model.py:60-110, H calls of predict_next_state_change :112-180) +
 * get_rewards_trajectory (setpoint_distance_reward_mapper.py:144-149) + the value of compute_mean_lcb_trajectory
 * (gp_mpc_controller.py:267-276) ], all H steps inside one launch.
 *   actions_dev (B,H,A) model-space actions in [0,1];  mu0_host (D), S0_host (D,D): initial state (same for all candidates);
 *   include_time / time0: ModelConfig.include_time_model, current_time_idx (gp_model.py:101-102)
 * Outputs (each nullable): mu_out_dev (B,H+1,D), Sig_out_dev (B,H+1,D,D) (index 0 = input state, :91-92), cost_mu_out_dev
 * (B,H+1) = -rewards, cost_var_out_dev (B,H+1), J_out_dev (B) = the objective the optimiser / argmin sees.  The cost outputs need
 * gpmpc_set_cost for this (D, A); with all three NULL the call is the plain predict_trajectory.
 * gpmpc_last_rollout_path: 0 = fused-horizon kernel (a workgroup per candidate), 1 = streaming kernel (per-point arrays beyond the
 * LDS), 2 = batch-major path (per step: workgroups own 128 x 128 tiles of beta beta^T - iK and loop over candidates; D <= 4, tables
 * beyond an XCD's L2, B >= 2 x CUs).  gpmpc_last_cluster: workgroups per candidate of the last fused-horizon launch; > 1 = the
 * few-candidate cooperative form -- the reference evaluates ONE sequence per objective call (restarts_optim 1-2,
 * gp_mpc_controller.py:125-141), so while candidates x cluster fit the chip a cluster of workgroups shares each candidate's
 * horizon step (bit-identical to one workgroup per candidate at equal "rows_per_chunk").
 */
int gpmpc_rollout(gpmpc_t* h, const double* actions_dev, const double* mu0_host, const double* S0_host, int B, int H, int A,
                  int include_time, double time0, double* mu_out_dev, double* Sig_out_dev, double* cost_mu_out_dev,
                  double* cost_var_out_dev, double* J_out_dev, void* stream);
int gpmpc_last_rollout_path(gpmpc_t* h);
int gpmpc_last_cluster(gpmpc_t* h);

/*
 * Objective AND analytic gradient: J_out_dev (B), grad_out_dev (B,H,A) = dJ/d(actions) -- what the reference obtains with
 * `mean_cost.backward()` (gp_mpc_controller.py:277) and hands to scipy as `jac` (:132-139, :285): forward rollout, pairwise
 * moment pass over the stored trajectory, reverse sweep.  clip_lower_bound_cost_to_0 clips the value only (pass-through clamp).
 * Other outputs as gpmpc_rollout (nullable).  Supported for D <= 8 with A (+ time) <= 6 and for 8 < D <= 16; otherwise
 * GPMPC_ERR_LIMIT (callers then difference gpmpc_rollout).  gpmpc_last_grad_path (bit mask, a test hook: which moment passes ran):
 * 1 separable off-diagonal pairs (matrix cores), 2 tile moments of the diagonal pairs, 4 streaming element-wise pass, 8 the
 * 8 < D <= 16 pass, 16 tile moments formed inside the batch-major forward, 32 mean part by its own kernel.
 */
int gpmpc_rollout_grad(gpmpc_t* h, const double* actions_dev, const double* mu0_host, const double* S0_host, int B, int H, int A,
                       int include_time, double time0, double* J_out_dev, double* grad_out_dev, double* mu_out_dev,
                       double* Sig_out_dev, double* cost_mu_out_dev, double* cost_var_out_dev, void* stream);
int gpmpc_last_grad_path(gpmpc_t* h);

/*
 * gpmpc_objective_grad_host  <->  ONE call of compute_mean_lcb_trajectory (gp_mpc_controller.py:229-285) as scipy's L-BFGS-B
 * makes it (:133-141: one action sequence per evaluation, host arrays in, (float, host gradient) out): gpmpc_rollout_grad for
 * B = 1 with host buffers on both sides; returns when the results are in *result_host (the last kernel raises a completion word
 * the call polls; `stream` itself may still be draining).  actions_host (H,A).  *result_host: pinned host
 * buffer owned by the handle, valid until the next call: J (1) | grad (H,A) | mu (H+1,D) | Sig (H+1,D,D) | cost_mu (H+1) |
 * cost_var (H+1) (trajectory and stage costs: what the reference caches for IterationInformation, :279-283).
 */
int gpmpc_objective_grad_host(gpmpc_t* h, const double* actions_host, const double* mu0_host, const double* S0_host,
                              int H, int A, int include_time, double time0, const double** result_host, void* stream);

/*
 * gpmpc_argmin  <->  the keep-the-best rule of gp_mpc_controller.py:146-148 on a vector of objectives: first strict minimum wins;
 * a NaN in GLOBAL slot 0 is adopted and never displaced; any other NaN is never selected.  `first_global_index` = global index of
 * J_dev[0] (the shard offset when candidates are sharded over GPUs); the returned index is global; nothing selectable: index -1,
 * J = +inf.  gpmpc_argmin synchronises and writes HOST memory; gpmpc_argmin_async does not: it writes the device record
 * out_dev[0] = best J, [1] = (double) global index, [2 .. 2+HA) = the winning sequence when actions_dev (B,HA) is given -- ready
 * to be all-gathered over RCCL as is (the ONE exchange of a sharded control step).
 */
int gpmpc_argmin(gpmpc_t* h, const double* J_dev, int B, long long first_global_index, double* best_J_host,
                 long long* best_idx_host, void* stream);
int gpmpc_argmin_async(gpmpc_t* h, const double* J_dev, int B, long long first_global_index, const double* actions_dev, int HA,
                       double* out_dev, void* stream);

/*
 * Candidate optimisation whose loop stays on the device (replaces B sequential scipy restarts, gp_mpc_controller.py:125-141, by a
 * cross-entropy search over the same box [0,1]^(H*A)): per iteration B vectors are drawn around the current mean / std
 * (iteration 0: uniform; slot 0 = the incumbent / `first_candidate_host`), mapped to model actions (mapper 0: identity reshape,
 * normalization_action_mapper.py:21-23; 1: scaled deltas + cumulative sum + pass-through clamp, derivative_action_mapper.py:28-35,
 * with max_change_host (A), action_prev_host (A)), evaluated by one rollout launch, and the n_elite best refit mean and std.
 * Nothing is read back between iterations.  best_out_dev (H*A + 1) = [best vector | its objective].  Draws: Philox4x32-10 keyed
 * by `seed`, or noise_dev (iterations, B, H*A) (iteration 0 uniforms, later standard normals: the parity tests' hook).  2 <= B <= 4096.
 *
 * Sharded over GPUs: gpmpc_cem_local = this GPU's slice [first, first + B_local) of iteration `iteration` (draws
 * indexed by the GLOBAL candidate, so the union of the slices is the single-GPU population) -> elites_out_dev (n_elite, 2 + H*A)
 * = [J | global index | vector], sorted, padded with (+inf, INT_MAX); the caller all-gathers them (RCCL); gpmpc_cem_merge refits
 * on the union (lists * n_elite <= 4096) -> state_dev = [mean | std | best vector | best J], bit for bit the single-GPU state.
 */
int gpmpc_cem_search(gpmpc_t* h, const double* mu0_host, const double* S0_host, int B, int H, int A, int include_time, double time0,
                     int iterations, int n_elite, unsigned long long seed, const double* first_candidate_host, int mapper,
                     const double* max_change_host, const double* action_prev_host, const double* noise_dev, double* best_out_dev,
                     void* stream);
int gpmpc_cem_local(gpmpc_t* h, const double* mu0_host, const double* S0_host, int B_total, int first, int B_local, int H, int A,
                    int include_time, double time0, int iteration, int n_elite, unsigned long long seed,
                    const double* first_candidate_host, int mapper, const double* max_change_host, const double* action_prev_host,
                    const double* noise_dev, const double* state_dev, double* elites_out_dev, void* stream);
int gpmpc_cem_merge(gpmpc_t* h, const double* elites_dev, int lists, int n_elite, int n, int iteration, double* state_dev,
                    void* stream);

/* bench.py: `reps` rollouts back to back bracketed by HIP events recorded on `stream`; average ms per launch in *ms_host. */
int gpmpc_rollout_timed(gpmpc_t* h, const double* actions_dev, const double* mu0_host, const double* S0_host, int B, int H, int A,
                        int include_time, double time0, double* J_out_dev, int reps, float* ms_host, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GPMPC_H */
