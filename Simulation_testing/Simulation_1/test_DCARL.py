"""Experiment 1 — online confidence-value estimation for 1 state x 11 candidate trajectories.

Drop-in for the reference script of the same path: same four module-level functions (same signatures), same
relative input files, same progress prints, same script-level globals and the same figure — but the loop
(reference lines 73-99) runs as hand-written HIP kernels on an MI355X through dcarl_amd.
Run from the repository root:  python Simulation_testing/Simulation_1/test_DCARL.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from dcarl_amd.params import Params  # noqa: E402
from dcarl_amd.reference_api import (CI_lower_bound, lower_bound, mean_value, run_simulation,  # noqa: E402,F401
                                     upper_bound)

if __name__ == "__main__":
    import matplotlib.pyplot as plt

    data = np.load('Simulation_testing/Simulation_1/data_carla.npy')
    true_action_values = np.load('Simulation_testing/Simulation_1/action_value_carla.npy')
    true_action_value = true_action_values[0]

    state_num, action_num = 1, 30                             # one state; 30 candidate slots, 11 of them ever sampled
    rule_act, n_thres = Params().rule_act, Params().n_thres   # 0 and 10: the estimator's defaults
    data_size, rate = 50000, 0.1                              # script globals of the reference the loop never reads

    g = run_simulation(data, true_action_values, state_num, action_num, limit=20000, log_every=2000)
    data_state_act = g["data_state_act"]                      # S1:41,80: every bucket's rewards in arrival order
    overall_value = g["overall_value"]                        # S1:48: declared, never filled by this script ([])
    TSRL_value = g["TSRL_value"]
    step_TSRL_value = g["step_TSRL_value"]
    step_TSRL_act = g["step_TSRL_act"]
    true_step_TSRL_value = g["true_step_TSRL_value"]
    activation_step = g["activation_step"]
    activation_value = g["activation_value"]
    state_data_len = g["state_data_len"]
    k = g["k"]

    plt.figure()
    id = 0
    print(activation_step[0])
    plt.plot(step_TSRL_value[id], color='black')
    plt.xlim((0, 20000))
    plt.show()
