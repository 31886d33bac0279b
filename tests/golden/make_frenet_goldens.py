"""Runs the UNMODIFIED reference `JunctionTrajectoryPlanner.calc_frenet_paths` (build container only) on 40 start
states and stores inputs + outputs as tests/golden/frenet_paths.npz."""
import os
import sys

import numpy as np

REF = "/root/reference/Simulation_testing/Simulation_Data_Collection/Data_From_Carla"
sys.path.insert(0, REF)
from Agent.zzz.JunctionTrajectoryPlanner import Frenet_state, JunctionTrajectoryPlanner  # noqa: E402

rng = np.random.RandomState(0)
planner = JunctionTrajectoryPlanner()
starts, trajs, costs = [], [], []
for i in range(40):
    st = Frenet_state()
    st.s0, st.c_d, st.c_d_d = rng.uniform(0, 120), rng.uniform(-3, 3), rng.uniform(-1, 1)
    st.c_d_dd = 0 if i % 2 == 0 else rng.uniform(-0.5, 0.5)      # JTP:272 sets 0; the commented branch a value
    c_speed = rng.uniform(0, 12)
    paths = planner.calc_frenet_paths(c_speed, st)
    starts.append([st.s0, c_speed, st.c_d, st.c_d_d, st.c_d_dd])
    trajs.append(np.array([[fp.d, fp.d_d, fp.d_dd, fp.d_ddd, fp.s, fp.s_d, fp.s_dd, fp.s_ddd] for fp in paths]))
    costs.append(np.array([[fp.cd, fp.cv, fp.cf] for fp in paths]))
t = np.array(paths[0].t)
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "frenet_paths.npz"), start=np.array(starts),
         traj=np.array(trajs), cost=np.array(costs), t=t)
print(np.array(trajs).shape, np.array(costs).shape, t)
