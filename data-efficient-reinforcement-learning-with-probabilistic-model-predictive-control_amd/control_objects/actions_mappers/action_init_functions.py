"""Candidate initialisers.  They draw from numpy's legacy GLOBAL generator exactly like the
reference (control_objects/actions_mappers/action_init_functions.py:4-10) so that a seeded
run evaluates the same candidate action sequences as the reference does."""
import numpy as np


def generate_mpc_action_init_random(len_horizon, dim_action):
    return np.random.uniform(low=0, high=1, size=(len_horizon, dim_action)).reshape(-1)


def generate_mpc_action_init_frompreviousiter(actions_mpc, dim_action):
    # shift the previous solution one step forward; the last step is repeated (in place, as the reference)
    actions_mpc[:-dim_action] = actions_mpc[dim_action:]
    return actions_mpc
