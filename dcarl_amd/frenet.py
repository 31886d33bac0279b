"""Candidate generation in the Frenet frame: host mirror of `JunctionTrajectoryPlanner.calc_frenet_paths`
(Simulation_testing/Simulation_Data_Collection/Data_From_Carla/Agent/zzz/JunctionTrajectoryPlanner.py:292-340, JTP below)
batched over many start states.  The module constants keep the reference's names (JTP:14-40).  No CPU fallback."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib

MAX_LEFT_WIDTH, MAX_RIGHT_WIDTH, D_ROAD_W = -4, 4, 2          # JTP:17-19
DT, MAXT, MINT = 0.3, 4.2, 4.0                                 # JTP:20-22
TARGET_SPEED, D_T_S, N_S_SAMPLE = 30.0 / 3.6, 15 / 3.6, 1      # JTP:23-25
KJ, KT, KD, KLAT, KLON = 0.1, 0.1, 1.0, 1.0, 1.0               # JTP:35-39
MAX_SPEED, MAX_ACCEL, MAX_CURVATURE = 50.0 / 3.6, 10.0, 500.0  # JTP:14-16
OBSTACLES_CONSIDERED, ROBOT_RADIUS, MOVE_GAP = 4, 1, 1         # JTP:28-31
GLOBAL_FIELDS = ("x", "y", "yaw", "ds", "c")
FIELDS = ("d", "d_d", "d_dd", "d_ddd", "s", "s_d", "s_dd", "s_ddd")


@dataclass
class FrenetCandidates:
    """traj[b, c, f, i]: field FIELDS[f] of candidate c of start state b at t[i]; cost[b, c] = (cd, cv, cf)."""
    traj: "object"
    cost: "object"
    t: np.ndarray
    offsets: np.ndarray       # lateral offset / horizon / target speed of candidate c
    horizons: np.ndarray
    speeds: np.ndarray

    def field(self, name):
        return self.traj[:, :, FIELDS.index(name), :]


class FrenetSampler:
    def __init__(self, target_speed=TARGET_SPEED, dts=D_T_S):
        self.device = _lib.require_gpu()
        self.target_speed, self.dts = target_speed, dts        # JTP:62-63
        d = np.arange(MAX_LEFT_WIDTH, MAX_RIGHT_WIDTH + 1, D_ROAD_W, dtype=np.float64)          # JTP:304
        T = np.arange(MINT, MAXT, DT)                                                            # JTP:307
        tv = np.arange(target_speed - dts * N_S_SAMPLE, target_speed + dts * N_S_SAMPLE, dts)   # JTP:318
        nt = [len(np.arange(0.0, Ti, DT)) for Ti in T]                                           # JTP:312
        if len(d) > 16 or len(T) > 8 or len(tv) > 8:
            raise ValueError("candidate grid larger than the C-ABI struct (16 offsets, 8 horizons, 8 speeds)")
        g = _lib.CFrenetGrid()
        g.n_d, g.n_T, g.n_v, g.nt_max = len(d), len(T), len(tv), max(nt)
        for i, x in enumerate(d): g.d[i] = x
        for i, x in enumerate(T): g.T[i] = x; g.nt[i] = nt[i]
        for i, x in enumerate(tv): g.tv[i] = x
        g.dt, g.target_speed = DT, target_speed
        g.kj, g.kt, g.kd, g.klat, g.klon = KJ, KT, KD, KLAT, KLON
        self.grid = g
        self.t = np.arange(0.0, T[int(np.argmax(nt))], DT)
        self.offsets = np.repeat(d, len(T) * len(tv))
        self.horizons = np.tile(np.repeat(T, len(tv)), len(d))
        self.speeds = np.tile(tv, len(d) * len(T))
        self.n_candidates = len(d) * len(T) * len(tv)

    def calc_frenet_paths(self, c_speed, s0, c_d, c_d_d, c_d_dd, want_traj=True, want_cost=True, out=None):
        """JTP:292-340 for B start states (arrays of equal length, or one (B, 5) tensor `{s0, c_speed, c_d, c_d_d,
        c_d_dd}` passed as `c_speed` with the other arguments None)."""
        import torch
        if s0 is None:
            start = c_speed.to(self.device, torch.float64).contiguous()
        else:
            start = torch.from_numpy(np.ascontiguousarray(
                np.stack(np.broadcast_arrays(*[np.asarray(x, np.float64) for x in (s0, c_speed, c_d, c_d_d, c_d_dd)]), -1)
                .reshape(-1, 5))).to(self.device)
        B, NC, NT = start.shape[0], self.n_candidates, self.grid.nt_max
        traj = cost = None
        if out is not None:
            traj, cost = out.traj, out.cost
        if want_traj and traj is None:
            traj = torch.empty((B, NC, 8, NT), dtype=torch.float64, device=self.device)
        if want_cost and cost is None:
            cost = torch.empty((B, NC, 3), dtype=torch.float64, device=self.device)
        _lib.check(_lib.load().dcarl_frenet_candidates_f64(_lib.ptr(start), B, C.byref(self.grid), _lib.ptr(traj if want_traj else None),
                                                          _lib.ptr(cost if want_cost else None), _lib.stream_ptr()),
                   "dcarl_frenet_candidates_f64")
        return FrenetCandidates(traj if want_traj else None, cost if want_cost else None, self.t, self.offsets,
                                self.horizons, self.speeds)


@dataclass
class GlobalPaths:
    """glob[b, c, f, i]: GLOBAL_FIELDS[f]; path_len[b, c] = samples inside the reference path's spline."""
    glob: "object"
    path_len: "object"

    def field(self, name):
        return self.glob[:, :, GLOBAL_FIELDS.index(name), :]


class ReferencePath:
    """The reference path as the reference's `Spline2D` (Agent/zzz/cubic_spline_planner.py): natural cubic splines
    x(s), y(s) over the cumulative chord length of the way points (`generate_target_course`, JTP:278-290).  The small
    tridiagonal solve happens once on the host, like in the reference."""

    def __init__(self, x, y, device):
        import torch
        x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
        s = np.concatenate([[0.0], np.cumsum(np.hypot(np.diff(x), np.diff(y)))])        # Spline2D.__calc_s

        def coeffs(v):                                   # natural cubic spline: c = 0 at both ends (Spline.__init__)
            n, h = len(s), np.diff(s)
            A = np.zeros((n, n))
            A[0, 0] = A[-1, -1] = 1.0
            r = np.arange(1, n - 1)
            A[r, r - 1], A[r, r], A[r, r + 1] = h[:-1], 2.0 * (h[:-1] + h[1:]), h[1:]
            rhs = np.zeros(n)
            rhs[1:-1] = 3.0 * (v[2:] - v[1:-1]) / h[1:] - 3.0 * (v[1:-1] - v[:-2]) / h[:-1]
            c = np.linalg.solve(A, rhs)
            d = (c[1:] - c[:-1]) / (3.0 * h)
            b = (v[1:] - v[:-1]) / h - h * (c[1:] + 2.0 * c[:-1]) / 3.0
            return v[:-1], b, c[:-1], d

        seg = np.stack(coeffs(x) + coeffs(y), axis=1)                                    # (n-1, 8)
        self.s = s
        self.knots = torch.from_numpy(s).to(device)
        self.segments = torch.from_numpy(np.ascontiguousarray(seg)).to(device)


def _limits():
    lim = _lib.CFrenetLimits()
    _lib.load().dcarl_frenet_default_limits(C.byref(lim))
    lim.n_predict = len(np.arange(0.0, MAXT, DT))                                       # predict.py:88
    return lim


def calc_global_paths(sampler: FrenetSampler, cands: FrenetCandidates, path: ReferencePath) -> GlobalPaths:
    """JTP:342-379 for every candidate of every start state."""
    import torch
    B, NC, _, NT = cands.traj.shape
    glob = torch.empty((B, NC, 5, NT), dtype=torch.float64, device=sampler.device)
    plen = torch.empty((B, NC), dtype=torch.int32, device=sampler.device)
    _lib.check(_lib.load().dcarl_frenet_global_paths_f64(_lib.ptr(cands.traj), B, C.byref(sampler.grid), _lib.ptr(path.knots),
                                                        _lib.ptr(path.segments), path.knots.numel(), _lib.ptr(glob),
                                                        _lib.ptr(plen), _lib.stream_ptr()), "dcarl_frenet_global_paths_f64")
    return GlobalPaths(glob, plen)


def get_optimal_trajectory(sampler: FrenetSampler, cands: FrenetCandidates, paths: GlobalPaths, obstacles=None,
                           want_flags=False):
    """JTP:123-130 per start state: index + 1 of the cheapest candidate that passes `check_paths` (JTP:381-394) and
    `predict.check_collision` (predict.py:21-60), 0 = brake.  obstacles: (B, n_obs, 5) `{x, y, vx, vy, yaw}` of the
    vehicles kept by `found_interested_vehicles` (see `nearest_vehicles`), or None."""
    import torch
    B, NC = paths.path_len.shape
    if obstacles is None:
        obs, n_obs = None, 0
    else:
        obs = torch.as_tensor(obstacles, dtype=torch.float64).to(sampler.device).contiguous()
        n_obs = obs.shape[1]
    choice = torch.empty(B, dtype=torch.int32, device=sampler.device)
    flags = torch.empty((B, NC), dtype=torch.uint8, device=sampler.device) if want_flags else None
    lim = _limits()
    _lib.check(_lib.load().dcarl_frenet_select(_lib.ptr(cands.traj), _lib.ptr(paths.glob), _lib.ptr(paths.path_len),
                                               _lib.ptr(cands.cost), _lib.ptr(obs), n_obs, B, C.byref(sampler.grid),
                                               C.byref(lim), _lib.ptr(choice), _lib.ptr(flags), _lib.stream_ptr()),
               "dcarl_frenet_select")
    return (choice, flags) if want_flags else choice


def nearest_vehicles(ego_xy, vehicles, k=OBSTACLES_CONSIDERED):
    """predict.py:62-82 found_interested_vehicles: the k vehicles nearest to the ego position (stable order).
    ego_xy (2,), vehicles (n, 5) -> (min(k, n), 5)."""
    vehicles = np.asarray(vehicles, np.float64).reshape(-1, 5)
    d = np.linalg.norm(vehicles[:, :2] - np.asarray(ego_xy, np.float64), axis=1)
    return vehicles[np.argsort(d, kind="stable")[:k]]
