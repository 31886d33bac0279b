"""Episode-return reduction on MI355X: the step that produces the cumulative-reward column.

* ``episode_returns`` — the CARLA collector's arithmetic: step reward ``0.1*sqrt(v)``, -100 on a collision, 0 when stuck
  (Test_Scenarios/TestScenario_Town03.py:402-421) summed per episode (Agent/drl_library/dqn/dqn_value_collect.py:119),
  plus the AveSpeed the collector logs.
* ``nstep_backup`` — the field vehicle's ``RLS.add_data`` value stream (stable_baselines/deepq/RLS.py:185-215): a 10-deep
  trajectory buffer; a transition leaves it with its own reward, the last ten of a finished episode get the terminal
  reward times ``gamma**k``."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

COLLISION, PASSED, STUCK = 1, 2, 4          # flag bits of a step (TS:406, 413, 418)


def _dev64(x, dev, dtype=torch.float64):
    return torch.as_tensor(x).to(device=dev, dtype=dtype).contiguous()


def episode_returns(vx, vy, flags, ep_off, want_steps: bool = True):
    """Returns (episode_reward f64 [E], ave_speed f64 [E], step_reward f64 [N] or None)."""
    dev = _lib.require_gpu()
    lib = _lib.load()
    vx, vy = _dev64(vx, dev), _dev64(vy, dev)
    flags = _dev64(flags, dev, torch.uint8)
    ep_off = _dev64(ep_off, dev, torch.int64)
    E, N = ep_off.numel() - 1, vx.numel()
    if vy.numel() != N or flags.numel() != N:
        raise ValueError("vx, vy and flags must have one entry per step")
    if E < 0 or (E > 0 and (int(ep_off[0]) != 0 or int(ep_off[-1]) != N or bool((ep_off[1:] < ep_off[:-1]).any()))):
        raise ValueError("ep_off must rise from 0 to the number of steps")
    ep = torch.empty(max(E, 0), dtype=torch.float64, device=dev)
    sp = torch.empty(max(E, 0), dtype=torch.float64, device=dev)
    st = torch.empty(N, dtype=torch.float64, device=dev) if want_steps else None
    _lib.check(lib.dcarl_episode_returns_f64(_lib.ptr(vx), _lib.ptr(vy), _lib.ptr(flags), _lib.ptr(ep_off), max(E, 0),
                                             _lib.ptr(st), _lib.ptr(ep), _lib.ptr(sp), _lib.stream_ptr()),
               "dcarl_episode_returns_f64")
    return ep, sp, st


def gamma_powers(gamma: float, horizon: int) -> np.ndarray:
    out = (C.c_double * max(horizon, 1))()
    _lib.load().dcarl_gamma_powers(float(gamma), int(horizon), out)
    return np.array(out[:horizon], dtype=np.float64)


def nstep_backup(rew, ep_off, ep_done, gamma: float = 0.95, horizon: int = 10):
    """RLS.add_data (RLS:185-215) for whole streams: returns (value f64 [N], recorded bool [N])."""
    dev = _lib.require_gpu()
    lib = _lib.load()
    rew = _dev64(rew, dev)
    ep_off = _dev64(ep_off, dev, torch.int64)
    ep_done = _dev64(ep_done, dev, torch.uint8)
    E, N = ep_off.numel() - 1, rew.numel()
    if ep_done.numel() != E:
        raise ValueError("ep_done must have one entry per episode")
    if E > 0 and (int(ep_off[0]) != 0 or int(ep_off[-1]) != N or bool((ep_off[1:] < ep_off[:-1]).any())):
        raise ValueError("ep_off must rise from 0 to the number of transitions")
    gp = torch.from_numpy(gamma_powers(gamma, horizon)).to(dev)
    value = torch.empty(N, dtype=torch.float64, device=dev)
    rec = torch.empty(N, dtype=torch.uint8, device=dev)
    _lib.check(lib.dcarl_nstep_backup_f64(_lib.ptr(rew), _lib.ptr(ep_off), _lib.ptr(ep_done), max(E, 0), _lib.ptr(gp),
                                          int(horizon), _lib.ptr(value), _lib.ptr(rec), _lib.stream_ptr()),
               "dcarl_nstep_backup_f64")
    return value, rec.bool()
