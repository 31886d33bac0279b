"""Experiment 2 — the same estimator for 20 states x 11 actions plus the cross-state overall_value.

Drop-in for the reference script of the same path (functions, input files, globals, figures); the loop
(reference lines 72-105) and the overall_value running sum run as HIP kernels on an MI355X through dcarl_amd.
Run from the repository root:  python Simulation_testing/Simulation_2/test_DCARL.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from dcarl_amd.params import Params  # noqa: E402
from dcarl_amd.reference_api import (CI_lower_bound, lower_bound, mean_value, plot_states,  # noqa: E402,F401
                                     run_simulation, upper_bound)

if __name__ == "__main__":
    import matplotlib.pyplot as plt

    data = np.load('Simulation_testing/Simulation_2/data.npy')
    true_action_values = np.load('Simulation_testing/Simulation_2/action_value.npy')
    true_action_value = true_action_values[0]

    state_num, action_num = true_action_values.shape          # 20 states x 11 candidate trajectories
    rule_act, n_thres = Params().rule_act, Params().n_thres   # 0 and 10: the estimator's defaults
    data_size, rate = 50000, 0.1                              # script globals of the reference the loop never reads

    g = run_simulation(data, true_action_values, state_num, action_num, limit=20000, with_overall=True)
    data_state_act = g["data_state_act"]                      # S2:41,80: every bucket's rewards in arrival order
    TSRL_value = g["TSRL_value"]
    step_TSRL_value = g["step_TSRL_value"]
    step_TSRL_act = g["step_TSRL_act"]
    true_step_TSRL_value = g["true_step_TSRL_value"]
    activation_step = g["activation_step"]
    activation_value = g["activation_value"]
    overall_value = g["overall_value"]
    state_data_len = g["state_data_len"]
    sorted_state_data_len = g["sorted_state_data_len"]
    k = g["k"]
    max_len = sorted_state_data_len[0][1]

    plot_states(g, plt=plt)
    plt.show()
