"""Expected quadratic cost of a Gaussian state (reference
control_objects/states_reward_mappers/setpoint_distance_reward_mapper.py).

Trajectory costs on the MPC hot path are computed inside the HIP rollout kernel; this host
class covers the O(D^2) single-state bookkeeping calls (logging in get_action /
compute_cost_unnormalized) and documents the formula the kernel implements.
"""
import math

import torch


def normal_cdf(x, mu, sigma):
    return 0.5 * (1 + torch.erf((x - mu) / (sigma * math.sqrt(2))))


class AbstractStateRewardMapper:
    def __init__(self, config):
        self.config = config


class SetpointStateRewardMapper(AbstractStateRewardMapper):
    def _quadratic(self, err, var, W):
        # E[e^T W e] and Var[e^T W e] for e ~ N(err, var)
        cost_mu = torch.trace(var @ W) + err @ W @ err
        TS = W @ var
        cost_var = torch.trace(2 * TS @ TS) + 4 * err @ TS @ W @ err
        return cost_mu, cost_var

    def get_reward(self, state_mu, state_var, action):
        """Stage cost of ONE state/action (reference :12-68, 1-D branch).  Returns (-cost_mu, cost_var)."""
        cfg = self.config
        err = torch.cat((state_mu, action), -1) - cfg.target_state_action_norm
        na = action.shape[0]
        full_var = torch.block_diag(state_var, torch.zeros((na, na), dtype=state_var.dtype))
        cost_mu, cost_var = self._quadratic(err, full_var, cfg.weight_matrix_cost)
        if cfg.use_constraints:
            # the reference hands the variance diagonal to a function expecting a std (:63-64); kept
            sd = state_var.diag()
            cost_mu = cost_mu + (1 - normal_cdf(cfg.state_max, state_mu, sd)).sum(-1) \
                + normal_cdf(cfg.state_min, state_mu, sd).sum(-1)
        return -cost_mu, cost_var

    def get_reward_terminal(self, state_mu, state_var):
        cfg = self.config
        cost_mu, cost_var = self._quadratic(state_mu - cfg.target_state_norm, state_var, cfg.weight_matrix_cost_terminal)
        return -cost_mu, cost_var

    def get_rewards_trajectory(self, states_mu, states_var, actions):
        """Host restatement for small cases / debugging; the controller uses the kernel's outputs."""
        rs, vs = [], []
        for t in range(actions.shape[0]):
            r, v = self.get_reward(states_mu[t], states_var[t], actions[t])
            rs.append(r)
            vs.append(v)
        r, v = self.get_reward_terminal(states_mu[-1], states_var[-1])
        rs.append(r)
        vs.append(v)
        return torch.stack(rs), torch.stack(vs)
