"""Monte-Carlo return sampler — drop-in for the reference script of the same path.

Same four functions (add_an_act_data, random_state_norm, random_state_manual, Data_Generation), same output
files at the same relative paths.  By default the draws come from the generators the reference itself draws from
(NumPy's global RandomState, Python's ``random``) in the reference's order, and the arithmetic runs as HIP kernels on an
MI355X (dcarl_amd.sampler): after ``np.random.seed(s); random.seed(s)`` the three files equal the reference's bit for
bit.  ``legacy_streams=False`` selects the library's own Philox-4x32-10 counter generator on the GPU instead (same
distributions, the fast path for large N).  Run from the repository root:
    python Simulation_testing/Simulation_Data_Collection/Data_Sampling/data_sampling.py
"""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "..")))
from dcarl_amd.reference_api import (Data_Generation, add_an_act_data, random_state_manual,  # noqa: E402,F401
                                     random_state_norm)

if __name__ == "__main__":
    Data_Generation()
