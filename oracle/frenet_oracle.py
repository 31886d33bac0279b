"""CPU restatement (NumPy, float64) of the reference's Frenet candidate generation -- TEST INFRASTRUCTURE ONLY.

Follows Simulation_testing/Simulation_Data_Collection/Data_From_Carla/Agent/zzz/JunctionTrajectoryPlanner.py:
quintic_polynomial JTP:397-446, quartic_polynomial JTP:448-491, calc_frenet_paths JTP:292-340, constants JTP:14-40.
PARITY PINNED: tests/golden/frenet_paths.npz holds the output of the unmodified reference function for 40 start
states (tests/golden/make_frenet_goldens.py); this restatement reproduces it to 1e-12."""
from __future__ import annotations

import numpy as np

MAX_LEFT_WIDTH, MAX_RIGHT_WIDTH, D_ROAD_W = -4, 4, 2
DT, MAXT, MINT = 0.3, 4.2, 4.0
TARGET_SPEED, D_T_S, N_S_SAMPLE = 30.0 / 3.6, 15 / 3.6, 1
KJ, KT, KD, KLAT, KLON = 0.1, 0.1, 1.0, 1.0, 1.0


def quintic(xs, vxs, axs, xe, vxe, axe, T):
    a0, a1, a2 = xs, vxs, axs / 2.0
    A = np.array([[T**3, T**4, T**5], [3 * T**2, 4 * T**3, 5 * T**4], [6 * T, 12 * T**2, 20 * T**3]])
    b = np.array([xe - a0 - a1 * T - a2 * T**2, vxe - a1 - 2 * a2 * T, axe - 2 * a2])
    x = np.linalg.solve(A, b)                                   # JTP:419
    return np.array([a0, a1, a2, x[0], x[1], x[2]])


def quartic(xs, vxs, axs, vxe, axe, T):
    a0, a1, a2 = xs, vxs, axs / 2.0
    A = np.array([[3 * T**2, 4 * T**3], [6 * T, 12 * T**2]])
    b = np.array([vxe - a1 - 2 * a2 * T, axe - 2 * a2])
    x = np.linalg.solve(A, b)                                   # JTP:466
    return np.array([a0, a1, a2, x[0], x[1], 0.0])


def derivs(a, t):
    """point, 1st, 2nd, 3rd derivative (JTP:425-446)."""
    return (a[0] + a[1] * t + a[2] * t**2 + a[3] * t**3 + a[4] * t**4 + a[5] * t**5,
            a[1] + 2 * a[2] * t + 3 * a[3] * t**2 + 4 * a[4] * t**3 + 5 * a[5] * t**4,
            2 * a[2] + 6 * a[3] * t + 12 * a[4] * t**2 + 20 * a[5] * t**3,
            6 * a[3] + 24 * a[4] * t + 60 * a[5] * t**2)


def calc_frenet_paths(c_speed, s0, c_d, c_d_d, c_d_dd, target_speed=TARGET_SPEED, dts=D_T_S):
    """One start state -> traj (n_cand, 8, nt), cost (n_cand, 3), in the reference's candidate order."""
    traj, cost = [], []
    for di in np.arange(MAX_LEFT_WIDTH, MAX_RIGHT_WIDTH + 1, D_ROAD_W):
        for Ti in np.arange(MINT, MAXT, DT):
            lat = quintic(c_d, c_d_d, c_d_dd, di, 0.0, 0.0, Ti)
            t = np.arange(0.0, Ti, DT)
            d = derivs(lat, t)
            for tv in np.arange(target_speed - dts * N_S_SAMPLE, target_speed + dts * N_S_SAMPLE, dts):
                lon = quartic(s0, c_speed, 0.0, tv, 0.0, Ti)
                s = derivs(lon, t)
                jp, js = np.sum(d[3] ** 2), np.sum(s[3] ** 2)
                ds = (target_speed - s[1][-1]) ** 2
                cd = KJ * jp + KT * Ti + KD * d[0][-1] ** 2
                cv = KJ * js + KT * Ti + KD * ds
                traj.append(np.stack(d + s))
                cost.append([cd, cv, KLAT * cd + KLON * cv])
    return np.stack(traj), np.array(cost)
