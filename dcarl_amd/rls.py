"""Field variant of the confidence test ("RLS"): host mirror of
Field_testing/Software_and_Raw_Data_on_Self-Driving_Vehicle/software/src/tools/DCARL/stable_baselines/deepq/RLS.py
for its statistics / test-time decision path (RLS:120-181), batched over many observations.

The reference keeps the visited (state, action) rows in an R-tree (third-party ``rtree``) and asks it, one query at a
time, which rows' boxes contain the query point; here the table lives in HBM and ``dcarl_rls_neighbour_stats_f64``
scans it for a whole batch of queries.  Names follow the reference (`visited_state_dist`, `visited_times_thres`,
`act_test`, `_calculate_visited_times`, `_calculate_statistics_index`).  No CPU fallback."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib

OBS_DIMENSION = 20                                            # RLS:30
# RLS:68 box half-widths: ego (x, y, vx, vy), four surrounding vehicles (x, y, vx, vy), action
VISITED_STATE_DIST = (1, 0.3, 2, 50, 10, 0.3, 2, 50, 10, 0.3, 2, 50, 10, 0.3, 2, 50, 10, 0.3, 2, 50, 0.1)


@dataclass(frozen=True)
class RlsParams:
    visited_times_thres: int = 30      # RLS:14
    min_rl_visits: int = 5             # RLS:141
    rule_mean_gate: float = -0.1       # RLS:141
    confidence_thres: float = 0.5      # RLS:120

    def to_c(self):
        return _lib.CRlsParams(self.visited_times_thres, self.min_rl_visits, self.rule_mean_gate, self.confidence_thres)


class RLS:
    """Visited-state table on the GPU + the reference's statistics and test-time policy.

    visited_state: (N, 21) rows `obs[20], action` (visited_state.txt); visited_value: (N, 2) rows `action, value`
    (visited_value.txt) or (N,) values."""

    def __init__(self, visited_state, visited_value, visited_times_thres=30, visited_state_dist=VISITED_STATE_DIST,
                 params: RlsParams | None = None, is_training: bool = False):
        import torch
        self.is_training = is_training                          # RLS:16,23: act() dispatches on it
        self.device = _lib.require_gpu()
        self.params = params or RlsParams(visited_times_thres=visited_times_thres)
        st = np.ascontiguousarray(np.asarray(visited_state, dtype=np.float64).reshape(-1, OBS_DIMENSION + 1))
        val = np.asarray(visited_value, dtype=np.float64)
        if val.ndim == 2:
            val = val[:, 1]                                     # RLS:172 value_array_av[:,1]
        if len(val) != len(st):
            raise ValueError(f"{len(st)} visited states but {len(val)} values")
        self.visited_state = torch.from_numpy(st).to(self.device)
        self.visited_state_value = torch.from_numpy(np.ascontiguousarray(val)).to(self.device)
        self.visited_state_dist = torch.tensor(visited_state_dist, dtype=torch.float64, device=self.device)
        if self.visited_state_dist.numel() != OBS_DIMENSION + 1:
            raise ValueError("visited_state_dist needs 21 entries")
        self.visited_state_counter = len(st)                    # RLS:52
        self._ws = None

    @staticmethod
    def state_with_action(obs, action):
        """RLS:100-102 for a batch: (B, 20) observations + one action id (scalar or (B,)) -> (B, 21) query points."""
        obs = np.asarray(obs, dtype=np.float64).reshape(-1, OBS_DIMENSION)
        a = np.broadcast_to(np.asarray(action, dtype=np.float64).reshape(-1, 1), (len(obs), 1))
        return np.concatenate([obs, a], axis=1)

    def statistics(self, queries):
        """count, mean, var for (Q, 21) query points = RLS:161-163 + RLS:165-181 per point (mean = var = -1 where
        count == 0; the reference also returns sigma = sqrt(var), which nothing downstream uses)."""
        import torch
        lib = _lib.load()
        q = queries if isinstance(queries, torch.Tensor) else torch.from_numpy(
            np.ascontiguousarray(np.asarray(queries, dtype=np.float64).reshape(-1, OBS_DIMENSION + 1)))
        q = q.to(self.device, torch.float64).contiguous()
        Q, N = q.shape[0], self.visited_state_counter
        need = lib.dcarl_rls_workspace_bytes(N, Q)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(max(need, 16), dtype=torch.uint8, device=self.device)
        count = torch.empty(Q, dtype=torch.int64, device=self.device)
        mean = torch.empty(Q, dtype=torch.float64, device=self.device)
        var = torch.empty(Q, dtype=torch.float64, device=self.device)
        _lib.check(lib.dcarl_rls_neighbour_stats_f64(_lib.ptr(self.visited_state), _lib.ptr(self.visited_state_value), N,
                                                     _lib.ptr(self.visited_state_dist), _lib.ptr(q), Q,
                                                     _lib.ptr(self._ws), _lib.ptr(count), _lib.ptr(mean), _lib.ptr(var),
                                                     _lib.stream_ptr()), "dcarl_rls_neighbour_stats_f64")
        return count, mean, var

    def _calculate_visited_times(self, obs_with_action):
        return self.statistics(obs_with_action)[0]

    def _calculate_statistics_index(self, obs_with_action):
        count, mean, var = self.statistics(obs_with_action)
        return mean, var, var.clamp_min(0).sqrt().where(count > 0, var)       # RLS:167-168 (-1, -1, -1) when unvisited

    def decide(self, count, mean, var, n_cand):
        """RLS:120-157 from statistics laid out [B][1 + n_cand] (column 0 = rule action)."""
        import torch
        lib = _lib.load()
        B = count.numel() // (1 + n_cand)
        action = torch.empty(B, dtype=torch.int32, device=self.device)
        cp = self.params.to_c()
        _lib.check(lib.dcarl_rls_decide(_lib.ptr(count.contiguous()), _lib.ptr(mean.contiguous()),
                                        _lib.ptr(var.contiguous()), B, n_cand, C.byref(cp), _lib.ptr(action),
                                        _lib.stream_ptr()), "dcarl_rls_decide")
        return action

    def act_test(self, obs, RL_action=None, candidates=range(1, 8)):
        """RLS:120-157 for a batch of observations (B, 20): the first candidate action whose neighbourhood mean beats
        the rule action's with the configured confidence, else 0.  (`RL_action` is unused, as in the reference.)"""
        cands = list(candidates)
        if cands != list(range(1, len(cands) + 1)):
            raise ValueError("candidates must be 1..n (the returned id is the column index)")
        obs = np.asarray(obs, dtype=np.float64).reshape(-1, OBS_DIMENSION)
        B = len(obs)
        q = np.stack([self.state_with_action(obs, a) for a in [0] + cands], axis=1).reshape(-1, OBS_DIMENSION + 1)
        count, mean, var = self.statistics(q)
        return self.decide(count, mean, var, len(cands))

    # ---- train-time policy (RLS:78-118) ------------------------------------------------------------------------------
    def _rule_statistics(self, obs):
        obs = np.asarray(obs, dtype=np.float64).reshape(-1, OBS_DIMENSION)
        count, mean, _ = self.statistics(self.state_with_action(obs, 0))
        return count, mean

    def _explore_draws(self, count, explore_motivation):
        """The exploration draws of RLS:112.  Injected (array of B values), or drawn like the reference does: Python's
        ``random.uniform(-1, 0)``, one draw per observation that passed the visit-count test, in order (seed-compatible)."""
        import random
        import torch
        B = count.numel()
        if explore_motivation is not None:
            e = np.broadcast_to(np.asarray(explore_motivation, dtype=np.float64).reshape(-1), (B,))
        else:
            enough = (count >= self.params.visited_times_thres).cpu().numpy()
            e = np.zeros(B)
            for b in np.flatnonzero(enough):                       # RLS:107-112: the draw happens only past the first test
                e[b] = random.uniform(-1, 0)
        return torch.from_numpy(np.array(e, dtype=np.float64)).to(self.device)

    def _gate(self, obs, RL_action, explore_motivation, want_action):
        import torch
        count, mean = self._rule_statistics(obs)
        B = count.numel()
        explore = self._explore_draws(count, explore_motivation)
        use = torch.empty(B, dtype=torch.uint8, device=self.device)
        act = rl = None
        if want_action:
            rl = torch.as_tensor(np.broadcast_to(np.asarray(RL_action, dtype=np.int32).reshape(-1), (B,)).copy()).to(self.device)
            act = torch.empty(B, dtype=torch.int32, device=self.device)
        _lib.check(_lib.load().dcarl_rls_gate_train(_lib.ptr(count), _lib.ptr(mean), _lib.ptr(explore), _lib.ptr(rl), B,
                                                    self.params.visited_times_thres, _lib.ptr(act), _lib.ptr(use),
                                                    _lib.stream_ptr()), "dcarl_rls_gate_train")
        return act, use

    def should_use_rule(self, obs, explore_motivation=None):
        """RLS:94-116 for a batch (B, 20): True where the rule action is not explored enough (fewer than
        ``visited_times_thres`` visits) or the exploration draw falls below its mean value.  -> bool tensor (B,)."""
        return self._gate(obs, None, explore_motivation, False)[1].bool()

    def act_train(self, obs, RL_action, explore_motivation=None):
        """RLS:85-89: 0 where ``should_use_rule``, else the DQN's (epsilon-greedy) action.  -> int32 tensor (B,)."""
        return self._gate(obs, RL_action, explore_motivation, True)[0]

    def act(self, obs, RL_action, explore_motivation=None):
        """RLS:78-82: ``act_train`` while training, ``act_test`` otherwise."""
        return self.act_train(obs, RL_action, explore_motivation) if self.is_training else self.act_test(obs, RL_action)
