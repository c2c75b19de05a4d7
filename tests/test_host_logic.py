"""CPU tests of the host-side mirror (no GPU, no compute calls into the HIP library): the library
loads and exports every symbol of include/gpmpc.h, config/mapper/memory logic, gradient chain rule."""
import os
import re

import numpy as np
import pytest
import torch

from helpers import load, workload_of

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import gp_mpc_amd
    from gp_mpc_amd import _lib
    header = open(os.path.join(ROOT, "include", "gpmpc.h")).read()
    declared = set(re.findall(r"\b(gpmpc_[a-z_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    lib = _lib.load()                      # raises if a bound symbol is missing
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.gpmpc_abi_version() == _lib.ABI_VERSION
    assert os.path.basename(gp_mpc_amd.LIB_PATH) == "libgpmpc_hip.so"


def test_header_is_plain_c_and_a_c_program_links_against_the_library(tmp_path):
    """The boundary is a C ABI: include/gpmpc.h must compile as C99 (not only as C++), and a C program that takes the
    address of every entry point must link against libgpmpc_hip.so and run (no GPU needed for gpmpc_abi_version)."""
    import shutil
    import subprocess
    import gp_mpc_amd
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    header = open(os.path.join(ROOT, "include", "gpmpc.h")).read()
    names = sorted(set(re.findall(r"\b(gpmpc_[a-z_]+)\s*\(", header)))
    src = tmp_path / "consumer.c"
    src.write_text(
        '#include <stdio.h>\n#include "gpmpc.h"\n'
        "typedef void (*entry_t)(void);\n"
        "int main(void) {\n"
        "    entry_t entry[] = {" + ", ".join(f"(entry_t){n}" for n in names) + "};\n"
        "    unsigned i, n = 0;\n"
        "    for (i = 0; i < sizeof entry / sizeof entry[0]; ++i) n += entry[i] != 0;\n"
        '    printf("%u %d\\n", n, gpmpc_abi_version());\n'
        "    return 0;\n}\n")
    libdir = os.path.dirname(gp_mpc_amd.LIB_PATH)
    exe = tmp_path / "consumer"
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
           "-L", libdir, "-lgpmpc_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    n, ver = r.stdout.split()
    from gp_mpc_amd import _lib
    assert int(n) == len(names) and int(ver) == _lib.ABI_VERSION


def test_no_cpu_fallback_without_gpu():
    import gp_mpc_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        gp_mpc_amd.HipEngine(0)


def test_product_package_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing in the shipped package may import or load it."""
    pkg = os.path.join(ROOT, "data-efficient-reinforcement-learning-with-probabilistic-model-predictive-control_amd")
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|oracle[./]", re.M)
    seen = 0
    for dirpath, _, files in os.walk(pkg):
        if "build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                seen += 1
                assert not pat.search(open(os.path.join(dirpath, f)).read()), (dirpath, f)
    assert seen > 10


def test_derivative_mapper_matches_reference_golden():
    g = load("lcb_grad_deriv")
    w = workload_of(g)
    from gp_mpc_amd.control_objects.actions_mappers.mappers import DerivativeActionMapper
    from gp_mpc_amd.config_classes import ActionsConfig
    N, D, A, E, H, B = w.dims
    m = DerivativeActionMapper(np.zeros(A), np.ones(A), H, ActionsConfig(True, list(g["max_change"])))
    m.action_model_previous_iter = torch.as_tensor(g["action_prev"])
    got = m.mpc_to_model_batch(w.actions.reshape(B, -1))
    assert np.allclose(got, g["actions_model"], rtol=0, atol=1e-15)
    one = m.transform_action_mpc_to_action_model(w.actions[1].reshape(-1)).numpy()
    assert np.array_equal(one, got[1])
    # chain rule with pass-through clamp: d model[t] / d u[s] = 2 m for s <= t
    gm = np.random.default_rng(0).standard_normal((H, A))
    want = np.array([[2 * 0.3 * gm[s:, a].sum() for a in range(A)] for s in range(H)]).reshape(-1)
    assert np.allclose(m.chain_grad_model_to_mpc(gm), want)


def test_normalization_mapper_and_bounds():
    from gp_mpc_amd.control_objects.actions_mappers.mappers import NormalizationActionMapper
    from gp_mpc_amd.config_classes import ActionsConfig
    m = NormalizationActionMapper(np.array([-2.0]), np.array([2.0]), 5, ActionsConfig())
    u = np.linspace(0, 1, 5)
    assert np.array_equal(m.transform_action_mpc_to_action_model(u).numpy(), u.reshape(5, 1))
    assert m.bounds == [(0, 1)] * 5
    raw = m.transform_action_model_to_action_raw(torch.as_tensor(u.reshape(5, 1)), update_internals=True)
    assert np.allclose(raw.numpy().ravel(), -2 + 4 * u) and m.n_iter_ctrl == 1
    assert np.allclose(m.transform_action_raw_to_action_model(np.array([0.0])).numpy(), [0.5])


def test_config_broadcasting_like_reference():
    from gp_mpc_amd.config_classes import ModelConfig
    mc = ModelConfig(gp_init={"noise_covar.noise": [1e-5] * 3, "base_kernel.lengthscale": [0.5, 0.6, 0.7],
                              "outputscale": [5e-2] * 3}, include_time_model=True, init_lengthscale_time=100)
    mc.extend_dimensions_params(dim_state=3, dim_input=5)
    ls = mc.gp_init["base_kernel.lengthscale"]
    assert ls.shape == (3, 5) and torch.all(ls[:, -1] == 100) and torch.all(ls[1, :-1] == 0.6)
    assert mc.min_lengthscale.shape == (3, 5) and mc.max_std_noise.shape == (3,)
    mc2 = ModelConfig(gp_init={"noise_covar.noise": [1e-5] * 2, "base_kernel.lengthscale": [0.5, 0.5],
                               "outputscale": [5e-2] * 2})
    mc2.extend_dimensions_params(dim_state=2, dim_input=3)
    assert mc2.gp_init["base_kernel.lengthscale"].shape == (2, 3)


def test_reward_config_matrices():
    from gp_mpc_amd.config_classes import RewardConfig
    rc = RewardConfig(target_state_norm=[1, 0.5], weight_state=[1, 0.1], weight_state_terminal=[5, 2],
                      target_action_norm=[0.5], weight_action=[1e-3])
    assert rc.weight_matrix_cost.shape == (3, 3) and rc.weight_matrix_cost[2, 2] == 1e-3
    assert torch.equal(rc.target_state_action_norm, torch.tensor([1, 0.5, 0.5], dtype=torch.float64))


def test_memory_admission_growth_and_dummy_point():
    from gp_mpc_amd.control_objects.memories.gp_memory import Memory
    from gp_mpc_amd.config_classes import MemoryConfig
    cfg = MemoryConfig(True, [1e-2, 1e-2], [1e-3, 1e-3], points_batch_memory=4)
    mem = Memory(cfg, dim_input=3, dim_state=2)
    x, y = mem.get()
    assert x.shape == (1, 3) and y.shape == (1, 2) and not x.any() and not y.any()      # dummy zero point
    s = torch.tensor([0.1, 0.2], dtype=torch.float64)
    a = torch.tensor([0.5], dtype=torch.float64)
    for k in range(11):                      # beyond points_batch_memory: the reference raises here
        pred = s + (0.05 if k % 2 == 0 else 0.0)
        mem.add(s, a, s + 0.01 * k, 0.0, iter_ctrl=k, predicted_state=pred, predicted_state_std=torch.tensor([0.1, 0.1]))
    mem.prepare_for_model()
    x, y = mem.get()
    assert mem.len_mem == 11 and len(x) == mem.len_mem_model == int(mem.active_data_mask[:11].sum())
    assert torch.allclose(y[0], mem.states_next[0] - s)
    assert mem.get_mask_model_inputs().shape == (11,)


@pytest.mark.parametrize("name", ["memory_trace", "memory_trace_time", "memory_trace_nocheck"])
@pytest.mark.parametrize("as_numpy", [False, True])
def test_memory_admission_rule_against_the_reference_trace(name, as_numpy):
    """The admission rule (reference gp_memory.py:48-63) and the model-memory bookkeeping (:66-111): the package's
    Memory replays the seeded stream the REFERENCE's Memory was run on (tools/gen_golden.py memory_trace_case) and must
    admit exactly the same points and hand `prepare_inference` exactly the same (x, y) after every `prepare_for_model`.
    `as_numpy`: run_env_function.py hands the predictions over as numpy arrays (iter_info.predicted_states[1])."""
    from gp_mpc_amd.control_objects.memories.gp_memory import Memory
    from gp_mpc_amd.config_classes import MemoryConfig
    g = load(name)
    D, A, it = int(g["D"]), int(g["A"]), bool(g["include_time"])
    n = len(g["states"])
    cfg = MemoryConfig(bool(g["check"]), list(g["thresholds_err"]), list(g["thresholds_std"]), points_batch_memory=16)   # grows twice
    mem = Memory(cfg, dim_input=D + A + int(it), dim_state=D, include_time_model=it)
    x0, y0 = mem.get()
    assert np.array_equal(x0.numpy(), g["empty_x"]) and np.array_equal(y0.numpy(), g["empty_y"])
    conv = (lambda v: np.asarray(v)) if as_numpy else (lambda v: torch.tensor(v, dtype=torch.float64))
    snaps = []
    t = lambda v: torch.tensor(v, dtype=torch.float64)
    for k in range(n):
        mem.add(t(g["states"][k]), t(g["actions"][k]), t(g["states_next"][k]), float(g["rewards"][k]), iter_ctrl=k,
                predicted_state=conv(g["predicted"][k]) if g["has_pred"][k] else None,
                predicted_state_std=conv(g["predicted_std"][k]) if g["has_std"][k] else None)
        if (k + 1) % int(g["prepare_every"]) == 0:
            mem.prepare_for_model()
            x, y = mem.get()
            snaps.append((mem.len_mem_model, x.numpy().copy(), y.numpy().copy()))
    assert np.array_equal(mem.active_data_mask[:n], g["admitted"])
    assert [s[0] for s in snaps] == list(g["snap_len"])
    assert np.array_equal(snaps[0][1], g["snap_first_x"]) and np.array_equal(snaps[0][2], g["snap_first_y"])
    assert np.array_equal(snaps[-1][1], g["final_x"]) and np.array_equal(snaps[-1][2], g["final_y"])
    assert np.array_equal(mem.inputs[:n].numpy(), g["inputs"]) and np.array_equal(mem.iter_ctrls[:n].numpy(), g["iter_ctrls"])
    if bool(g["check"]):
        assert np.array_equal(mem.errors[:n].numpy(), g["errors"], equal_nan=True)
        assert np.array_equal(mem.stds[:n].numpy(), g["stds"], equal_nan=True)
    xt, yt = mem.get_memory_total()
    assert np.array_equal(xt.numpy(), g["total_x"]) and np.array_equal(yt.numpy(), g["total_y"])
    assert np.array_equal(mem.get_mask_model_inputs(), g["mask_model_inputs"])
    assert mem.len_mem == int(g["len_mem"]) and mem.len_mem_last_processed == int(g["len_mem_last_processed"])


def test_single_state_reward_matches_golden_rows():
    """Host get_reward / get_reward_terminal (logging path) vs the reference's trajectory rewards."""
    g = load("traj_c1")
    w = workload_of(g)
    from gp_mpc_amd.control_objects.states_reward_mappers.setpoint_distance_reward_mapper import SetpointStateRewardMapper
    from gp_mpc_amd.config_classes import RewardConfig
    D = 3
    rc = RewardConfig(target_state_norm=list(w.target[:D]), weight_state=list(np.diag(w.W)[:D]),
                      weight_state_terminal=list(np.diag(w.W_T)), target_action_norm=list(w.target[D:]),
                      weight_action=list(np.diag(w.W)[D:]), exploration_factor=w.kappa)
    rm = SetpointStateRewardMapper(rc)
    mu, Sig, act = torch.as_tensor(g["mu"][0]), torch.as_tensor(g["Sig"][0]), torch.as_tensor(w.actions[0])
    r, v = rm.get_rewards_trajectory(mu, Sig, act)
    assert np.allclose(r.numpy(), g["rewards"][0], rtol=1e-12, atol=1e-15)
    assert np.allclose(v.numpy(), g["reward_vars"][0], rtol=1e-10, atol=1e-18)


def test_hyperparameter_holder_contract():
    from gp_mpc_amd.control_objects.models.gp_model import GpHyperParameters
    h = GpHyperParameters([0.5, 0.6], 0.05, 1e-5)
    assert h.covar_module.base_kernel.lengthscale.shape == (1, 2) and h.likelihood.noise.shape == (1,)
    h.initialize(**{"covar_module.base_kernel.lengthscale": np.array([[1.0, 2.0]]), "covar_module.outputscale": np.array(0.1),
                    "likelihood.noise": np.array([1e-4])})
    assert float(h.covar_module.outputscale) == 0.1 and h.covar_module.base_kernel.lengthscale[0, 1] == 2.0
    assert set(h.state_dict()) == set(GpHyperParameters.KEYS)


def test_training_improves_marginal_likelihood_cpu():
    """The optimiser loop of GpStateTransitionModel.train (restart inside the box, LBFGS, keep the best) with the
    oracle's torch expression standing in for gpmpc_mll (the product evaluates the loss on the GPU only)."""
    import queue
    from oracle.gp_training import neg_mll_torch
    from gp_mpc_amd.control_objects.models.gp_model import GpStateTransitionModel, SavedState, GpHyperParameters
    rng = np.random.default_rng(0)
    X = rng.uniform(size=(40, 2))
    Y = (0.3 * np.sin(4 * X[:, :1]) + 0.01 * rng.standard_normal((40, 1)))
    p0 = GpHyperParameters([5.0, 5.0], 0.9, 0.09).state_dict()
    st = SavedState(X, Y, [p0], {"min_lengthscale": np.full((1, 2), 4e-3), "max_lengthscale": np.full((1, 2), 10.0),
                                 "min_outputscale": np.array([1e-3]), "max_outputscale": np.array([0.95]),
                                 "min_std_noise": np.array([1e-3]), "max_std_noise": np.array([3e-1])})
    st.to_arrays()
    q = queue.Queue()
    torch.manual_seed(0)
    GpStateTransitionModel.train(q, st, 1e-1, 10, 1e-3, loss_evaluator=neg_mll_torch)
    (out,) = q.get()
    assert out["covar_module.base_kernel.lengthscale"].shape == (1, 2) and out["likelihood.noise"].shape == (1,)
    assert 4e-3 <= out["covar_module.base_kernel.lengthscale"].min() and out["likelihood.noise"][0] <= 0.09 + 1e-12


def test_training_always_answers_the_queue():
    """a training child that cannot create its engine (no GPU here) must still put exactly one result --
    the incoming hyper-parameters -- on the queue, so check_and_close_processes never blocks on a dead child."""
    import queue
    from gp_mpc_amd.control_objects.models.gp_model import GpStateTransitionModel, SavedState, GpHyperParameters
    p0 = GpHyperParameters([0.7, 0.8], 0.05, 1e-4).state_dict()
    st = SavedState(np.zeros((5, 2)), np.zeros((5, 1)), [p0],
                    {"min_lengthscale": np.full((1, 2), 4e-3), "max_lengthscale": np.full((1, 2), 10.0),
                     "min_outputscale": np.array([1e-3]), "max_outputscale": np.array([0.95]),
                     "min_std_noise": np.array([1e-3]), "max_std_noise": np.array([3e-1])})
    st.to_arrays()
    from gp_mpc_amd.control_objects.models.gp_model import TrainingFailed
    for dev in ("hip", "cpu"):                 # no GPU in this container / a device the product does not have
        q = queue.Queue()
        GpStateTransitionModel.train(q, st, 1e-1, 3, 1e-3, device=dev)
        answer = q.get_nowait()
        assert isinstance(answer, TrainingFailed) and answer.reason       # the parent can tell a failed training
        (out,) = answer
        assert np.array_equal(out["covar_module.base_kernel.lengthscale"], np.asarray(p0["covar_module.base_kernel.lengthscale"]))
        assert q.empty()
    # the marker survives the trip through a multiprocessing queue (pickle)
    import pickle
    back = pickle.loads(pickle.dumps(answer))
    assert isinstance(back, TrainingFailed) and back.reason == answer.reason and len(back) == 1
    # ... and a device the package cannot compute on is refused in the parent, at configuration time
    from gp_mpc_amd.config_classes import TrainingConfig
    with pytest.raises(ValueError):
        TrainingConfig(device="cpu")


def test_training_draws_its_restarts_in_the_reference_order():
    """Reference gp_model.py:236-252: per GP the restart is drawn outputscale, lengthscale, noise -- GP after GP.  With the
    per-GP searches running as lockstep threads the draws are made up front in that order, so a seeded run consumes the
    generator exactly like the sequential loop."""
    import queue
    from gp_mpc_amd.control_objects.models.gp_model import GpStateTransitionModel, SavedState, GpHyperParameters
    rng = np.random.default_rng(1)
    X = rng.uniform(size=(30, 2))
    Y = np.stack([np.sin(3 * X[:, 0]), np.cos(2 * X[:, 1])], axis=1) * 0.2
    cons = {"min_lengthscale": np.full((2, 2), 4e-3), "max_lengthscale": np.full((2, 2), 10.0),
            "min_outputscale": np.full(2, 1e-3), "max_outputscale": np.full(2, 0.95),
            "min_std_noise": np.full(2, 1e-3), "max_std_noise": np.full(2, 3e-1)}
    seen = []

    def spy(Xt, y, ls, osc, nz):
        seen.append((ls.detach().numpy().copy(), float(osc), float(nz)))
        from oracle.gp_training import neg_mll_torch
        return neg_mll_torch(Xt, y, ls, osc, nz)
    st = SavedState(X, Y, [GpHyperParameters([5.0, 5.0], 0.9, 0.09).state_dict() for _ in range(2)], cons)
    st.to_arrays()
    torch.manual_seed(3)
    GpStateTransitionModel.train(queue.Queue(), st, 1e-1, 1, 1e-3, loss_evaluator=spy)
    torch.manual_seed(3)
    want = []
    for a in range(2):
        r_os, r_ls, r_nz = torch.rand((), dtype=torch.float64), torch.rand(2, dtype=torch.float64), torch.rand((), dtype=torch.float64)
        want.append((4e-3 + (10.0 - 4e-3) * r_ls.numpy(), 1e-3 + (0.95 - 1e-3) * float(r_os), 1e-6 + (9e-2 - 1e-6) * float(r_nz)))
    # evaluation order without a device: GP 0 (incoming parameters, then its restart), then GP 1
    first_restart = [s for s in seen if not np.allclose(s[0], 5.0)]
    got0 = first_restart[0]
    got1 = next(s for s in first_restart if not np.allclose(s[0], got0[0]) and abs(s[1] - got0[1]) > 1e-12 and np.allclose(s[0], want[1][0], rtol=1e-9))
    assert np.allclose(got0[0], want[0][0], rtol=1e-9) and abs(got0[1] - want[0][1]) < 1e-9 and abs(got0[2] - want[0][2]) < 1e-9
    assert abs(got1[1] - want[1][1]) < 1e-9 and abs(got1[2] - want[1][2]) < 1e-9


def test_set_cost_follows_in_place_edits_of_the_reward_config():
    """the reference reads its reward config on every evaluation; the engine's copy is keyed on content."""
    from gp_mpc_amd.control_objects.models.gp_model import GpStateTransitionModel
    from gp_mpc_amd.config_classes import ModelConfig, RewardConfig

    class Eng:
        def __init__(self):
            self.sent = []

        def set_cost(self, *a):
            self.sent.append(a)
    eng = Eng()
    m = GpStateTransitionModel(ModelConfig(), 3, 1, engine=eng)
    rc = RewardConfig(target_state_norm=[0.5] * 3, weight_state=[1.0] * 3, weight_state_terminal=[1.0] * 3,
                      target_action_norm=[0.5], weight_action=[0.1])
    m.set_cost(rc)
    m.set_cost(rc)
    assert len(eng.sent) == 1
    rc.exploration_factor = 2.5
    m.set_cost(rc)
    assert len(eng.sent) == 2 and eng.sent[-1][3] == 2.5
    rc.target_state_action_norm[0] = 0.25
    m.set_cost(rc)
    assert len(eng.sent) == 3


def test_lockstep_restarts_drive_scipy_exactly_like_the_sequential_loop():
    """candidate_optimizer="lbfgs" host logic on a stand-in objective (no GPU): every restart's scipy L-BFGS-B solve gets,
    from the batched evaluation rounds, exactly the values a sequential loop would hand it, so end points and winner are
    identical; a restart that finishes early drops out of the rounds; an evaluation error reaches the caller."""
    from scipy.optimize import minimize
    import gp_mpc_amd  # noqa: F401
    from gp_mpc_amd.config_classes import (Config, ControllerConfig, ActionsConfig, RewardConfig, ObservationConfig,
                                           MemoryConfig, ModelConfig, TrainingConfig)
    from gp_mpc_amd import GpMpcController
    D, A, H, restarts = 3, 1, 6, 5                  # ModelConfig defaults are sized for 3 states + 1 action
    cfg = Config(observation_config=ObservationConfig(obs_var_norm=[1e-6] * D),
                 reward_config=RewardConfig(target_state_norm=[0.5] * D, weight_state=[1.0] * D, weight_state_terminal=[1.0] * D,
                                            target_action_norm=[0.5] * A, weight_action=[0.1] * A),
                 actions_config=ActionsConfig(limit_action_change=False, max_change_action_norm=[0.3] * A),
                 model_config=ModelConfig(), memory_config=MemoryConfig(points_batch_memory=16),
                 training_config=TrainingConfig(training_frequency=10 ** 9),
                 controller_config=ControllerConfig(len_horizon=H, restarts_optim=restarts, candidate_optimizer="lbfgs",
                                                    init_from_previous_actions=False))

    class Stub(GpMpcController):                      # the objective lives on the host; nothing touches the engine
        rounds = []

        def objective_and_gradient_batch(self, X, obs_mu, obs_var):
            X = np.asarray(X)
            self.rounds.append(X.shape[0])
            c = np.linspace(0.2, 0.9, X.shape[1])
            J = ((X - c) ** 2).sum(1) + 0.1 * np.sin(5 * X).sum(1)
            return J, 2 * (X - c) + 0.5 * np.cos(5 * X)

        def evaluate_candidates(self, actions_mpc_batch, obs_mu, obs_var, trajectories=False):
            return None

        def _cache_trajectory(self, out, idx):
            pass

        def _prepare(self):
            pass

    c = Stub(np.zeros(D), np.ones(D), np.zeros(A), np.ones(A), cfg, engine=object())
    np.random.seed(3)
    best = c._get_optimal_actions(None, None)
    np.random.seed(3)
    x0s = [np.random.uniform(0, 1, H * A) for _ in range(restarts)]

    def f(x):
        J, G = Stub.objective_and_gradient_batch(c, x[None], None, None)
        return float(J[0]), G[0]
    seq = [minimize(fun=f, x0=x0, jac=True, method="L-BFGS-B", bounds=c.actions_mapper.bounds,
                    options=cfg.controller.actions_optimizer_params) for x0 in x0s]
    assert np.array_equal(c.candidates_final_J, np.array([r.fun for r in seq]))
    win = int(np.argmin([r.fun for r in seq]))
    assert c.best_candidate_index == win and np.array_equal(c.actions_mpc_previous_iter, seq[win].x)
    assert best.shape == (H, A)
    assert c.lbfgs_evaluations == max(r.nfev for r in seq)               # rounds = the longest restart
    assert Stub.rounds[0] == restarts and min(Stub.rounds[:c.lbfgs_evaluations]) < restarts   # early finishers drop out

    class Broken(Stub):
        def objective_and_gradient_batch(self, X, obs_mu, obs_var):
            raise ValueError("device lost")
    b = Broken(np.zeros(D), np.ones(D), np.zeros(A), np.ones(A), cfg, engine=object())
    with pytest.raises(RuntimeError, match="batched evaluation failed"):
        b._get_optimal_actions(None, None)


def test_pending_best_combines_rank_records_like_the_reference_loop():
    """sharding.PendingBest.result on fabricated per-rank records [J, global index, winning actions]: lowest J wins, ties go
    to the lowest global index, a rank with nothing selectable (-1) is skipped, NaN in global slot 0 is adopted
    (gp_mpc_controller.py:146-148 applied across ranks)."""
    import gp_mpc_amd  # noqa: F401
    from gp_mpc_amd.sharding import PendingBest
    H, A = 2, 1

    def rec(J, idx, a):
        return [J, float(idx), a, a + 0.5]
    host = torch.tensor([rec(0.7, 3, 1.0), rec(0.4, 9, 2.0), rec(0.4, 6, 3.0)], dtype=torch.float64).reshape(-1)
    J, i, act = PendingBest(host, None, 3, H, A).result()
    assert (J, i) == (0.4, 6) and act.reshape(-1).tolist() == [3.0, 3.5]
    host = torch.tensor([rec(float("inf"), -1, 0.0), rec(0.9, 5, 2.0)], dtype=torch.float64).reshape(-1)
    assert PendingBest(host, None, 2, H, A).result()[:2] == (0.9, 5)
    host = torch.tensor([rec(float("nan"), 0, 7.0), rec(0.1, 5, 2.0)], dtype=torch.float64).reshape(-1)
    J, i, act = PendingBest(host, None, 2, H, A).result()
    assert J != J and i == 0 and act.reshape(-1).tolist() == [7.0, 7.5]
    host = torch.tensor([rec(float("inf"), -1, 0.0)], dtype=torch.float64).reshape(-1)
    with pytest.raises(FloatingPointError):
        PendingBest(host, None, 1, H, A).result()


def test_committed_bench_line_has_the_contract_fields():
    """profiles/r02b_c2_bench.json is the line bench.py printed on the MI355X: schema of the driver's contract."""
    import json
    d = json.loads(open(os.path.join(ROOT, "profiles", "r02b_c2_bench.json")).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "MPC trajectory rollouts/sec" and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(d["value"] - d["config"]["B_total"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference")
    assert d["scaling"] == "weak" and 0.0 < r["valu_busy_frac"] < 1.0 and r["peak_measured_fma_loop"] > 30.0
    d5 = json.loads(open(os.path.join(ROOT, "profiles", "r02b_c5_bench_B256.json")).read().strip().splitlines()[-1])
    assert d5["scaling"] == "strong" and d5["cpu_baseline"].get("extrapolated") is True and d5["config"]["N"] == 4096
    # the round-5 lines: median of five timed windows, and the reversed-batch bitwise check at the bench's own batch size
    for name, nwin in (("r05y_c2_bench.json", 5), ("r05y_c2_bench_v2.json", 5), ("r05y_c5_bench_B256.json", 1),
                       ("r06z_c2_bench.json", 5), ("r06z_c3_bench.json", 5), ("r06z_c4_bench.json", 1), ("r06z_c5_bench_B256.json", 1)):
        e = json.loads(open(os.path.join(ROOT, "profiles", name)).read().strip().splitlines()[-1])
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "windows"):
            assert k in e, (name, k)
        w = e["windows"]
        assert w["n"] == nwin == len(w["ms_per_step"]) and sorted(w["ms_per_step"])[nwin // 2] == e["ms_per_step"]
        assert abs(e["value"] - e["config"]["B_total"] / (e["ms_per_step"] * 1e-3)) / e["value"] < 1e-9
        assert e["parity"]["batch_independence"]["bitwise_equal_reversed_batch"] is True
        assert e["roofline"]["counters_note"] is None and 0.0 < e["roofline"]["valu_busy_frac"] < 1.0
        if name.startswith("r06z"):
            # round 6: the bounded fraction beside the contract's `frac` (which exceeds 1 on configs 3 / 4), the library path, the first window
            assert 0.0 < e["roofline"]["formulation"]["frac_formulation"] < 1.0
            assert e["config"]["lib_path"].endswith("libgpmpc_hip.so") and e["ms_per_step_first_window"] == w["ms_per_step"][0]
    # few-candidate latency lines (the reference's regime): one candidate per GPU, the cooperative form where the shape has one
    for name, coop in (("r06z_c2_B1_bench.json", True), ("r06z_c3_B1_bench.json", True), ("r06z_c1_B1_bench.json", False)):
        e = json.loads(open(os.path.join(ROOT, "profiles", name)).read().strip().splitlines()[-1])
        assert e["config"]["B_per_gpu"] == 1 and (e["config"]["workgroups_per_candidate"] > 1) == coop
        assert 0.0 < e["gradient"]["host_in_host_out_ms_per_evaluation"] < 1.0


def test_counter_derived_fields_of_the_stored_lines(tmp_path):
    """The counter-derived fields of a bench line are a pure function of the kernel time and profiles/pmc_*.json
    (bench.counter_figures): `bench.py --refresh-line` on a copy of the stored config-5 line reproduces them, the pipe occupancy
    prices a matrix instruction at 64 cycles and the executed flops at 256 multiply-adds per SQ_INSTS_VALU_MFMA_MOPS_F64 count
    (profiles/r04z_mfma_mops_unit.txt: 4 counts per v_mfma_f64_16x16x4_f64) -- nothing may execute more than the peak."""
    import json
    import shutil
    import subprocess
    import sys
    # (the lines of the LATEST evidence run: profiles/pmc_*.json hold one entry per workload key, stamped with that build)
    for name in ("r06z_c5_bench_B256.json", "r06z_c2_bench.json", "r06z_c3_bench.json", "r06z_c2_B4096_bench.json", "r06z_c4_bench.json"):
        src = os.path.join(ROOT, "profiles", name)
        stored = json.loads(open(src).read().strip().splitlines()[-1])
        cp = tmp_path / name
        shutil.copy(src, cp)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--refresh-line", str(cp)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        got = json.load(open(cp))
        a, b = stored["roofline"], got["roofline"]
        assert b["counters_note"] is None and b["build_id"] == a["build_id"] == b["counters"]["_build_id"]
        assert 0.3 < b["valu_busy_frac"] < 1.0 and 0.1 < b["executed"]["frac_of_peak"] < 1.0
        for k in ("kernel_ms", "frac", "achieved"):
            assert a[k] == b[k]
        assert stored["value"] == got["value"] and stored["gradient"] == got["gradient"]
        if "counters_refreshed" in a:             # the stored line already went through the refresh: idempotent
            assert abs(a["valu_busy_frac"] - b["valu_busy_frac"]) < 1e-12 and a["executed"] == b["executed"]
        # the work of the formulation the kernels execute: bounded by the peak on every shape, and by the issued fp64 flops
        f = b["formulation"]
        assert 0.0 < f["frac_formulation"] < 1.0 and f["frac_formulation"] <= b["executed"]["frac_of_peak"] * 1.0001
        if name.startswith("r06z_c5"):
            c = b["counters"]
            n_mfma = 0.25 * c["SQ_INSTS_VALU_MFMA_MOPS_F64"]
            want = ((c["SQ_INSTS_VALU"] - n_mfma) * 4 + n_mfma * 64) / (1024 * b["kernel_ms"] * 1e-3 * 2.4e9)
            assert abs(b["valu_busy_frac"] - want) < 1e-12 and 0.7 < want < 0.8
        else:                                     # no matrix instructions in the fused-horizon / batch-major forward: unchanged by the unit
            assert abs(a["valu_busy_frac"] - b["valu_busy_frac"]) < 1e-12
            assert abs(a["executed"]["frac_of_peak"] - b["executed"]["frac_of_peak"]) < 1e-12


def test_lockstep_training_batches_the_gps_and_isolates_a_failing_one():
    """GpStateTransitionModel.train with a device loss: the D per-GP LBFGS searches run as threads and every round of
    pending evaluations is ONE engine.mll call over the waiting GPs (reference: GP after GP, gp_model.py:233-290).  A fake
    engine (the oracle's closed form behind the engine's mll signature) stands in for the GPU here: call count, batch
    widths, the per-GP fallback when a batched call raises, and the result against the sequential CPU restatement."""
    import queue
    from oracle import gp_training
    import gp_mpc_amd.control_objects.models.gp_model as gm

    class FakeEngine:
        device = torch.device("cpu")
        calls = []

        def __init__(self, dev=None):
            pass

        def mll(self, X, Y, ls, osc, nz):
            X, Y = np.asarray(X), np.asarray(Y)
            ls, osc, nz = np.asarray(ls), np.asarray(osc).reshape(-1), np.asarray(nz).reshape(-1)
            FakeEngine.calls.append(Y.shape[1])
            if Y.shape[1] > 1 and len(FakeEngine.calls) == 3:
                raise RuntimeError("batched call fails once: the pending GPs must be evaluated one by one")
            out = {"loss": [], "d_lengthscale": [], "d_outputscale": [], "d_noise": []}
            for k in range(Y.shape[1]):
                loss, g_ls, g_os, g_nz = gp_training.neg_mll_and_grad(X, Y[:, k], ls[k], osc[k], nz[k])
                out["loss"].append(loss); out["d_lengthscale"].append(g_ls); out["d_outputscale"].append(g_os); out["d_noise"].append(g_nz)
            return {k: np.asarray(v) for k, v in out.items()}

        def close(self):
            pass

    rng = np.random.default_rng(2)
    X = rng.uniform(size=(30, 2))
    Y = np.stack([0.3 * np.sin(4 * X[:, 0]), 0.2 * np.cos(3 * X[:, 1]), 0.1 * X[:, 0] * X[:, 1]], axis=1) + 0.01 * rng.standard_normal((30, 3))
    cons = {"min_lengthscale": np.full((3, 2), 4e-3), "max_lengthscale": np.full((3, 2), 10.0),
            "min_outputscale": np.full(3, 1e-3), "max_outputscale": np.full(3, 0.95),
            "min_std_noise": np.full(3, 1e-3), "max_std_noise": np.full(3, 3e-1)}
    st = gm.SavedState(X, Y, [gm.GpHyperParameters([5.0, 5.0], 0.9, 0.09).state_dict() for _ in range(3)], dict(cons))
    st.to_arrays()
    import gp_mpc_amd.engine as engine_module
    real = engine_module.HipEngine
    engine_module.HipEngine = FakeEngine
    try:
        q = queue.Queue()
        torch.manual_seed(0)
        gm.GpStateTransitionModel.train(q, st, 1e-1, 4, 1e-3, device="hip")
        got = q.get_nowait()
    finally:
        engine_module.HipEngine = real
    assert not isinstance(got, gm.TrainingFailed)
    assert FakeEngine.calls[0] == 3 and max(FakeEngine.calls) == 3          # all GPs in one call while all are running
    assert FakeEngine.calls[3:6] == [1, 1, 1]                                # the failed batched call, GP by GP
    want, _ = gp_training.train_loop(X, Y, [{"lengthscale": [5.0, 5.0], "outputscale": 0.9, "noise": 0.09}] * 3, cons, 1e-1, 4, seed=0)
    for a in range(3):
        assert np.allclose(np.asarray(got[a][gm.GpHyperParameters.KEYS[0]]).ravel(), want[a]["lengthscale"], rtol=1e-5)
        assert abs(float(got[a][gm.GpHyperParameters.KEYS[1]]) - want[a]["outputscale"]) < 1e-5 * want[a]["outputscale"]


def test_lockstep_evaluation_that_times_out_withdraws_its_request():
    """a lockstep evaluation that gives up after WAIT_LIMIT must not leave its entry in `pending` -- `finished` of a
    sibling would otherwise count it, flush a batch that contains the stale request and leave a result nobody pops."""
    import threading
    import gp_mpc_amd.control_objects.models.gp_model as gm

    class Eng:
        calls = []

        def mll(self, X, Y, ls, osc, nz):
            Eng.calls.append(int(Y.shape[1]))
            n = Y.shape[1]
            return {"loss": np.zeros(n), "d_lengthscale": np.zeros((n, 2)), "d_outputscale": np.zeros(n), "d_noise": np.zeros(n)}

    shared = gm._LockstepMll(Eng(), torch.zeros(4, 2), torch.zeros(4, 2), 2)
    shared.WAIT_LIMIT = 0.0                      # first 5 s wait that is not served -> TimeoutError
    real_wait = shared.cond.wait
    shared.cond.wait = lambda timeout=None: real_wait(timeout=0.01) and False
    with pytest.raises(TimeoutError):
        shared.evaluate(0, torch.ones(2), torch.tensor(1.0), torch.tensor(0.1))
    assert shared.pending == {}
    shared.finished(0)                           # GP 0's thread ends (its LBFGS raised): one GP left running
    assert Eng.calls == [] and shared.results == {}
    shared.cond.wait = real_wait
    out = shared.evaluate(1, torch.ones(2), torch.tensor(1.0), torch.tensor(0.1))     # served alone, at once
    assert Eng.calls == [1] and out[0] == 0.0 and shared.results == {}


def _moment_items_by_lane(N, CH, NC, diag):
    """The lane mapping of pair_moments_kernel (grad_kernels.h: the work-item loop), lane by lane: rows of every work item."""
    NCU = (N + NC - 1) // NC
    RC = (N + CH - 1) // CH
    wpp = (RC * NCU + 63) // 64
    tri, run = [], 0
    for r in range(RC + 1):
        tri.append(run)
        first = (r * CH) // NC
        run += NCU - first if first < NCU else 0
    rows = []
    for slot in range(wpp):
        longest = 0
        for lane in range(64):
            flat = slot * 64 + lane
            if diag:
                if flat >= tri[RC]:
                    continue
                r = max(k for k in range(RC) if tri[k] <= flat)
                jc = (r * CH) // NC + (flat - tri[r])
            else:
                if flat >= RC * NCU:
                    continue
                r, jc = divmod(flat, NCU)
            jl = min(NC * jc + NC - 1, N - 1)
            i0 = r * CH
            i1 = min(i0 + CH, N)
            if diag:
                i1 = min(i1, jl + 1)
            longest = max(longest, i1 - i0)
        rows.append((longest + 3) & ~3)
    return rows


def test_moment_pass_schedule_model(tmp_path):
    """csrc/moment_schedule.h (host-only C++): its work items are those of the kernel's lane mapping, the chosen chunk length is
    admissible and never worse than the 64-row chunks it replaces, and the small memories get short chunks (config 1 had ONE item
    per pair with 64-row chunks: 3 of 16 wavefronts busy)."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    csrc = os.path.join(ROOT, "data-efficient-reinforcement-learning-with-probabilistic-model-predictive-control_amd", "csrc")
    src = tmp_path / "model.cc"
    src.write_text(r'''
#include "moment_schedule.h"
#include <cstdio>
#include <cstdlib>
using namespace gpmpc_hip;
int main(int argc, char** argv) {
    const int N = atoi(argv[1]), NC = atoi(argv[2]), NW = atoi(argv[3]), G = atoi(argv[4]), nd = atoi(argv[5]), no = atoi(argv[6]), CH = atoi(argv[7]);
    std::vector<int> pairs;
    for (int i = 0; i < nd; ++i) pairs.push_back(1);
    for (int i = 0; i < no; ++i) pairs.push_back(0);
    const int CH0 = N >= 64 ? 64 : ((N + 3) & ~3);
    const int want = choose_moment_chunk(N, NC, NW, CH0, pairs, [&](int, int& Gc, int& gz) { Gc = G; gz = 1; return true; });
    printf("%d %.6f %.6f\n", want, moment_schedule_cost(moment_items(N, NC, want), pairs, G, 1, NW), moment_schedule_cost(moment_items(N, NC, CH0), pairs, G, 1, NW));
    const MomentItems it = moment_items(N, NC, CH);
    for (int v : it.diag) printf("%d ", v);
    printf("\n");
    for (int v : it.full) printf("%d ", v);
    printf("\n");
    return 0;
}
''')
    exe = tmp_path / "model"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I", csrc, str(src), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    chosen = {}
    for (N, NC, NW, G, nd, no, CH) in [(50, 1, 16, 3, 3, 0, 8), (200, 1, 16, 3, 3, 0, 40), (200, 1, 16, 6, 3, 3, 52), (500, 1, 16, 2, 2, 0, 56),
                                       (200, 1, 8, 4, 4, 0, 64), (37, 2, 8, 3, 2, 1, 12), (3, 1, 16, 1, 1, 0, 4), (129, 1, 16, 3, 3, 0, 20)]:
        r = subprocess.run([str(exe)] + [str(v) for v in (N, NC, NW, G, nd, no, CH)], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stderr
        head, diag, full = r.stdout.strip("\n").split("\n")
        want, cost, cost0 = head.split()
        want = int(want)
        CH0 = 64 if N >= 64 else (N + 3) & ~3
        assert want % 4 == 0 and (want == CH0 or 8 <= want <= CH0)
        assert float(cost) <= float(cost0) * (1 + 1e-12)
        assert [int(v) for v in diag.split()] == _moment_items_by_lane(N, CH, NC, True)
        assert [int(v) for v in full.split()] == _moment_items_by_lane(N, CH, NC, False)
        chosen[(N, NW, nd, no)] = want
    assert chosen[(50, 16, 3, 0)] <= 20           # config 1: several items per pair instead of one
    assert 36 <= chosen[(200, 16, 3, 0)] <= 48    # config 2 with the off-diagonal pairs on the matrix cores: 30 items for 2 rounds of 16 wavefronts


def test_formulation_work_is_bounded_by_the_reference_count_and_by_the_executed_flops():
    """bench.formulation_work (the `roofline.formulation` block): the useful work of the evaluation forms the kernels choose per
    (pair, step) -- Taylor degree from the kernels' own bound, triangle-only diagonal pairs, separable off-diagonal pairs.  It must
    sit below the reference formulation's count (what `roofline.frac` prices) and below the fp64 flops the SIMDs
    actually issued for the same launch (SQ counters of the round-5 config-2 line), so that its fraction of the peak is <= 1
    by construction where `frac` (1.17 / 2.04 on configs 3 / 4) is not."""
    import json
    import bench
    from oracle import synth, gpmpc_oracle as orc
    for name in ("c1", "c2"):
        n, d, a, h, b, tm = synth.SHAPES[name]
        w = synth.make_workload(n, d, a, h, 2, include_time=tm, seed=0)
        ref = orc.evaluate_candidates(orc.Factors(w.X, w.Y, w.lengthscales, w.outputscales, w.noises), w)
        fw = bench.formulation_work(w.X, w.lengthscales, ref["mu"], ref["Sig"], a, d + a + (1 if tm else 0), 0)
        total = fw["flops_per_rollout"] + fw["exps_per_rollout"]
        assert 0.0 < total < bench.algorithmic_flops_per_rollout(n, d, a, d + a + (1 if tm else 0), h)
        assert abs(sum(fw["pair_steps_by_form"].values()) - 1.0) < 1e-12 and 1.0 <= fw["mean_taylor_degree"] <= 14.0
        if name == "c2":
            stored = json.loads(open(os.path.join(ROOT, "profiles", "r05y_c2_bench.json")).read().strip().splitlines()[-1])
            executed = stored["roofline"]["executed"]["fp64_flops_per_launch"]
            assert total * stored["config"]["B_per_gpu"] < executed            # useful work <= issued work
            assert fw["pair_steps_by_form"]["separable"] == 0.5               # the three off-diagonal pairs of D = 3, every step
