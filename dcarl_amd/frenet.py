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
