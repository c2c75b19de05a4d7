"""run_env -- the episode loop of the reference (rl_gp_mpc/run_env_function.py:15-50) without the
matplotlib/imageio visualisation side process (out of scope): same controller calls in the same
order, returns the per-step costs."""
import numpy as np

from .control_objects.controllers.gp_mpc_controller import GpMpcController


def run_env(env, control_config, visu_config=None, random_actions_init=10, num_steps=150, verbose=True,
            engine=None, device=None):
    ctrl = GpMpcController(observation_low=env.observation_space.low, observation_high=env.observation_space.high,
                           action_low=env.action_space.low, action_high=env.action_space.high,
                           config=control_config, engine=engine, device=device)
    obs = env.reset()
    costs = []
    for idx_ctrl in range(num_steps):
        action = ctrl.get_action(obs_mu=obs, random=(idx_ctrl < random_actions_init))
        iter_info = ctrl.get_iter_info()
        cost, cost_var = ctrl.compute_cost_unnormalized(obs, action)
        costs.append(cost)
        obs_new, reward, done, info = env.step(action)
        ctrl.add_memory(obs=obs, action=action, obs_new=obs_new, reward=-cost,
                        predicted_state=iter_info.predicted_states[1],
                        predicted_state_std=iter_info.predicted_states_std[1])
        obs = obs_new
        if verbose:
            print(str(iter_info))
    if hasattr(ctrl, "p_train") and not ctrl.p_train._closed:
        ctrl.p_train.join()                       # do not orphan a training process at episode end
    ctrl.check_and_close_processes()
    if hasattr(env, "__exit__"):
        env.__exit__()
    return np.array(costs), ctrl
