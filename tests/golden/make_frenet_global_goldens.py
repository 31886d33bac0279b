"""Runs the UNMODIFIED reference planner pieces (build container only): generate_target_course -> calc_frenet_paths ->
calc_global_paths -> get_optimal_trajectory with a `predict` object built on a stand-in dynamic map (plain attribute
holders for ego_vehicle / vehicles: the real one comes from CARLA) -> tests/golden/frenet_global.npz."""
import os
import sys
from types import SimpleNamespace as NS

import numpy as np

REF = "/root/reference/Simulation_testing/Simulation_Data_Collection/Data_From_Carla"
sys.path.insert(0, REF)
from Agent.zzz.JunctionTrajectoryPlanner import (DT, MAXT, MOVE_GAP, OBSTACLES_CONSIDERED, RADIUS_SPEED_RATIO,  # noqa: E402
                                                 Frenet_state, JunctionTrajectoryPlanner)
from Agent.zzz.predict import predict  # noqa: E402

rng = np.random.RandomState(1)
wx = np.array([0.0, 12.0, 25.0, 40.0, 52.0, 60.0, 75.0, 95.0, 120.0, 150.0])
wy = np.array([0.0, 1.0, 4.0, 9.0, 12.0, 11.0, 6.0, 2.0, 1.0, 0.5])
planner = JunctionTrajectoryPlanner()
_, _, _, _, csp = planner.generate_target_course(wx, wy)
starts, glob, plen, choice, vehicles_all = [], [], [], [], []
for i in range(60):
    st = Frenet_state()
    st.s0 = rng.uniform(0, csp.s[-1] - (3 if i % 10 == 9 else 45))     # every tenth start runs off the end of the path
    st.c_d, st.c_d_d, st.c_d_dd = rng.uniform(-3, 3), rng.uniform(-1, 1), 0
    c_speed = rng.uniform(0, 12)
    fplist = planner.calc_global_paths(planner.calc_frenet_paths(c_speed, st), csp)
    ex, ey = csp.calc_position(st.s0)
    n_veh = rng.randint(0, 7)
    veh = [NS(x=ex + rng.uniform(-5, 40), y=ey + rng.uniform(-6, 6), vx=rng.uniform(-3, 8), vy=rng.uniform(-2, 2),
              yaw=rng.uniform(-3, 3)) for _ in range(n_veh)]
    dm = NS(ego_vehicle=NS(x=ex, y=ey, v=c_speed), vehicles=veh)
    planner.obs_prediction = predict(dm, OBSTACLES_CONSIDERED, MAXT, DT, planner.radius, RADIUS_SPEED_RATIO, MOVE_GAP, c_speed)
    kept = np.full((OBSTACLES_CONSIDERED, 5), np.nan)
    ds = sorted([(np.linalg.norm([v.x - ex, v.y - ey]), j) for j, v in enumerate(veh)], key=lambda t: t[0])
    for r, (_, j) in enumerate(ds[:OBSTACLES_CONSIDERED]):
        kept[r] = [veh[j].x, veh[j].y, veh[j].vx, veh[j].vy, veh[j].yaw]
    tuples = [[fp, fp.cf, j] for j, fp in enumerate(fplist)]
    choice.append(planner.get_optimal_trajectory(tuples))
    g = np.zeros((len(fplist), 5, 14))
    for j, fp in enumerate(fplist):
        for f, v in enumerate((fp.x, fp.y, fp.yaw, fp.ds, fp.c)):
            g[j, f, :len(v)] = v
    glob.append(g)
    plen.append([len(fp.x) for fp in fplist])
    starts.append([st.s0, c_speed, st.c_d, st.c_d_d, st.c_d_dd, ex, ey])
    vehicles_all.append(kept)
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "frenet_global.npz"), wx=wx, wy=wy, start=np.array(starts),
         glob=np.array(glob), path_len=np.array(plen, np.int32), choice=np.array(choice, np.int32),
         vehicles=np.array(vehicles_all), knots=np.array(csp.s))
print(np.array(glob).shape, np.bincount(choice), np.array(plen).min())
