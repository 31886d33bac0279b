"""Monte-Carlo return sampler — drop-in for the reference script of the same path.

Same four functions (add_an_act_data, random_state_norm, random_state_manual, Data_Generation), same output
files at the same relative paths; the draws come from a Philox-4x32-10 counter generator running as HIP kernels
on an MI355X (dcarl_amd.sampler).  Run from the repository root:
    python Simulation_testing/Simulation_Data_Collection/Data_Sampling/data_sampling.py
"""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "..")))
from dcarl_amd.reference_api import (Data_Generation, add_an_act_data, random_state_manual,  # noqa: E402,F401
                                     random_state_norm)

if __name__ == "__main__":
    Data_Generation()
