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


# ---- global frame + screening: cubic_spline_planner.py Spline / Spline2D, JTP:342-394, predict.py:21-60,84-110 ------
import bisect
import math

MAX_SPEED, MAX_ACCEL, MAX_CURVATURE = 50.0 / 3.6, 10.0, 500.0
ROBOT_RADIUS, MOVE_GAP = 1, 1


def natural_spline_coefficients(x, a):
    """Per-segment coefficients b, c, d of the natural cubic spline through (x_k, a_k): the tridiagonal system of
    cubic_spline_planner.py Spline.__init__ / __calc_A / __calc_B (c = 0 at both ends), written with array slices."""
    x, a = np.asarray(x, np.float64), np.asarray(a, np.float64)
    n, h = len(x), np.diff(x)
    A = np.zeros((n, n))
    A[0, 0] = A[-1, -1] = 1.0
    r = np.arange(1, n - 1)
    A[r, r - 1], A[r, r], A[r, r + 1] = h[:-1], 2.0 * (h[:-1] + h[1:]), h[1:]
    B = np.zeros(n)
    B[1:-1] = 3.0 * (a[2:] - a[1:-1]) / h[1:] - 3.0 * (a[1:-1] - a[:-2]) / h[:-1]
    c = np.linalg.solve(A, B)
    d = (c[1:] - c[:-1]) / (3.0 * h)
    b = (a[1:] - a[:-1]) / h - h * (c[1:] + 2.0 * c[:-1]) / 3.0
    return b, c, d


class Spline:
    def __init__(self, x, y):
        self.x, self.a = list(x), list(y)
        self.b, self.c, self.d = natural_spline_coefficients(x, y)

    def calc(self, t, order=0):
        if t < self.x[0] or t > self.x[-1]:
            return None
        i = bisect.bisect(self.x, t) - 1
        dx = t - self.x[i]
        if order == 0:
            return self.a[i] + self.b[i] * dx + self.c[i] * dx ** 2.0 + self.d[i] * dx ** 3.0
        return self.b[i] + 2.0 * self.c[i] * dx + 3.0 * self.d[i] * dx ** 2.0


class Spline2D:
    def __init__(self, x, y):
        self.s = [0] + list(np.cumsum([math.sqrt(a ** 2 + b ** 2) for a, b in zip(np.diff(x), np.diff(y))]))
        self.sx, self.sy = Spline(self.s, x), Spline(self.s, y)


def calc_global_path(d, s, csp):
    """JTP:345-377 for one candidate: x, y, yaw, ds, c."""
    x, y = [], []
    for i in range(len(s)):
        ix, iy = csp.sx.calc(s[i]), csp.sy.calc(s[i])
        if ix is None:
            break
        iyaw = math.atan2(csp.sy.calc(s[i], 1), csp.sx.calc(s[i], 1))
        x.append(ix + d[i] * math.cos(iyaw + math.pi / 2.0))
        y.append(iy + d[i] * math.sin(iyaw + math.pi / 2.0))
    dx, dy = np.diff(np.array(x)), np.diff(np.array(y))
    yaw, ds = np.arctan2(dy, dx).tolist(), np.sqrt(dx ** 2 + dy ** 2).tolist()
    if yaw:
        yaw.append(yaw[-1]); ds.append(ds[-1])
    else:
        yaw.append(0.1); ds.append(0.1)
    c = []
    for i in range(len(yaw) - 1):
        if ds[i] < 0.00001:
            ds[i] = 0.1
        c.append((yaw[i + 1] - yaw[i]) / ds[i])
    return x, y, yaw, ds, c


def check_collision(x, nt, vehicles, y, n_predict, dt=DT):
    """predict.py:21-60 with the constant-velocity paths of predict.py:84-110 (front, back circle per vehicle)."""
    if len(vehicles) == 0 or nt < 2:
        return True
    for v in vehicles:
        for sign in (1.0, -1.0):
            for t in range(2, min(len(x) - 1, n_predict - 1), 2):
                px = v[0] + t * dt * v[2] + sign * math.cos(v[4]) * MOVE_GAP
                py = v[1] + t * dt * v[3] + sign * math.sin(v[4]) * MOVE_GAP
                if (px - x[t]) ** 2 + (py - y[t]) ** 2 <= ROBOT_RADIUS ** 2:
                    return False
    return True


def get_optimal_trajectory(traj, cost, csp, vehicles):
    """JTP:123-130 for one start state: traj (n_cand, 8, nt), cost (n_cand, 3) -> index + 1 or 0."""
    n_predict = len(np.arange(0.0, MAXT, DT))
    order = sorted(range(len(cost)), key=lambda i: cost[i][2])
    for i in order:
        s_d, s_dd = traj[i][5], traj[i][6]
        x, y, yaw, ds, c = calc_global_path(traj[i][0], traj[i][4], csp)
        if any(v > MAX_SPEED for v in s_d) or any(abs(a) > MAX_ACCEL for a in s_dd) or any(abs(k) > MAX_CURVATURE for k in c):
            continue
        if check_collision(x, traj.shape[2], vehicles, y, n_predict):
            return i + 1
    return 0
