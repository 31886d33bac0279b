"""Frenet candidate generation: oracle vs goldens produced by the unmodified reference function
(CPU) and the HIP kernels vs both (GPU)."""
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import frenet_oracle as fo  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden", "frenet_paths.npz")


def rel(a, b):
    return np.abs(a - b) / np.maximum(1.0, np.abs(b))


def test_oracle_reproduces_the_reference_function():
    g = np.load(GOLD)
    assert g["traj"].shape == (40, 10, 8, 14) and np.allclose(g["t"], np.arange(0.0, 4.0, 0.3))
    for i, (s0, c_speed, c_d, c_d_d, c_d_dd) in enumerate(g["start"]):
        traj, cost = fo.calc_frenet_paths(c_speed, s0, c_d, c_d_d, c_d_dd)
        assert rel(traj, g["traj"][i]).max() <= 1e-12 and rel(cost, g["cost"][i]).max() <= 1e-12


@pytest.mark.gpu
def test_candidates_vs_reference_goldens():
    import torch
    import dcarl_amd as dc
    g = np.load(GOLD)
    fs = dc.frenet.FrenetSampler()
    assert fs.n_candidates == 10 and np.array_equal(fs.t, g["t"])
    res = fs.calc_frenet_paths(torch.from_numpy(g["start"]), None, None, None, None)
    # closed-form 3x3 / 2x2 solves instead of np.linalg.solve: a few ulp of the coefficients, amplified by t^5 <= 900
    assert rel(res.traj.cpu().numpy(), g["traj"]).max() <= 1e-11
    assert rel(res.cost.cpu().numpy(), g["cost"]).max() <= 1e-11
    assert np.allclose(res.field("d")[:, :, -1].cpu().numpy()[:, ::2], res.field("d")[:, :, -1].cpu().numpy()[:, 1::2])
    s = g["start"]
    res2 = fs.calc_frenet_paths(s[:, 1], s[:, 0], s[:, 2], s[:, 3], s[:, 4], want_traj=False)      # array form, costs only
    assert res2.traj is None and torch.equal(res2.cost, res.cost)


@pytest.mark.gpu
def test_candidates_vs_oracle_other_speeds_and_sizes():
    import dcarl_amd as dc
    rng = np.random.RandomState(3)
    fs = dc.frenet.FrenetSampler(target_speed=9.0, dts=2.5)
    B = 3000
    start = np.column_stack([rng.uniform(0, 500, B), rng.uniform(0, 15, B), rng.uniform(-4, 4, B), rng.uniform(-2, 2, B),
                             rng.uniform(-1, 1, B)])
    res = fs.calc_frenet_paths(start[:, 1], start[:, 0], start[:, 2], start[:, 3], start[:, 4])
    traj, cost = res.traj.cpu().numpy(), res.cost.cpu().numpy()
    for b in rng.randint(0, B, 25):
        rt, rc = fo.calc_frenet_paths(start[b, 1], start[b, 0], start[b, 2], start[b, 3], start[b, 4], 9.0, 2.5)
        assert rel(traj[b], rt).max() <= 1e-11 and rel(cost[b], rc).max() <= 1e-11
    # boundary conditions hold for every candidate: start state reproduced at t = 0
    assert np.allclose(traj[:, :, 0, 0], start[:, None, 2]) and np.allclose(traj[:, :, 4, 0], start[:, None, 0])
    assert np.allclose(traj[:, :, 5, 0], start[:, None, 1]) and np.allclose(traj[:, :, 6, 0], 0.0)
    empty = fs.calc_frenet_paths(np.zeros(0), np.zeros(0), np.zeros(0), np.zeros(0), np.zeros(0))
    assert empty.traj.shape == (0, 10, 8, 14)


GOLD_G = os.path.join(REPO, "tests", "golden", "frenet_global.npz")


def _golden_vehicles(g, i):
    v = g["vehicles"][i]
    return v[~np.isnan(v[:, 0])]


def test_oracle_global_paths_and_choice_reproduce_the_reference():
    g = np.load(GOLD_G)
    csp = fo.Spline2D(g["wx"], g["wy"])
    assert np.allclose(csp.s, g["knots"], rtol=0, atol=1e-12)
    for i, st in enumerate(g["start"]):
        traj, cost = fo.calc_frenet_paths(st[1], st[0], st[2], st[3], st[4])
        for c in range(10):
            x, y, yaw, ds, cv = fo.calc_global_path(traj[c][0], traj[c][4], csp)
            n = g["path_len"][i, c]
            assert len(x) == n
            for f, v in enumerate((x, y, yaw, ds, cv)):
                assert rel(np.array(v), g["glob"][i, c, f, :len(v)]).max() <= 1e-10
        assert fo.get_optimal_trajectory(traj, cost, csp, _golden_vehicles(g, i)) == g["choice"][i]


@pytest.mark.gpu
def test_global_paths_and_choice_vs_reference_goldens():
    import torch
    import dcarl_amd as dc
    from dcarl_amd import frenet as fr
    g = np.load(GOLD_G)
    fs = fr.FrenetSampler()
    path = fr.ReferencePath(g["wx"], g["wy"], fs.device)
    assert np.allclose(path.s, g["knots"], rtol=0, atol=1e-12)
    st = g["start"]
    cands = fs.calc_frenet_paths(torch.from_numpy(np.ascontiguousarray(st[:, :5])), None, None, None, None)
    gp = fr.calc_global_paths(fs, cands, path)
    assert np.array_equal(gp.path_len.cpu().numpy(), g["path_len"])
    got, ref = gp.glob.cpu().numpy(), g["glob"]
    for f, tol in enumerate((1e-9, 1e-9, 1e-8, 1e-9, 1e-7)):        # x, y, yaw, ds, c (a difference quotient of yaw)
        assert rel(got[:, :, f], ref[:, :, f]).max() <= tol, fr.GLOBAL_FIELDS[f]
    # selection: every start state has its own obstacle count in the golden, the kernel takes a fixed n_obs per call
    for n_obs in range(5):
        idx = [i for i in range(len(st)) if len(_golden_vehicles(g, i)) == n_obs]
        if not idx:
            continue
        sub_c = fr.FrenetCandidates(cands.traj[idx].contiguous(), cands.cost[idx].contiguous(), cands.t, cands.offsets,
                                    cands.horizons, cands.speeds)
        sub_g = fr.GlobalPaths(gp.glob[idx].contiguous(), gp.path_len[idx].contiguous())
        obs = np.stack([_golden_vehicles(g, i) for i in idx]) if n_obs else None
        choice, flags = fr.get_optimal_trajectory(fs, sub_c, sub_g, obs, want_flags=True)
        assert choice.cpu().tolist() == g["choice"][idx].tolist()
        assert flags.shape == (len(idx), 10)
    assert len(set(g["choice"].tolist())) >= 4
    near = fr.nearest_vehicles([0.0, 0.0], [[9, 0, 0, 0, 0], [1, 1, 0, 0, 0], [3, 0, 0, 0, 0], [1, -1, 0, 0, 0], [5, 5, 0, 0, 0]])
    assert near[:, 0].tolist() == [1.0, 1.0, 3.0, 5.0]             # predict.py:62-82: nearest four, stable


@pytest.mark.gpu
def test_selection_edge_cases_vs_oracle():
    """Start states at / past the end of the reference path (paths cut short, down to no sample at all), speeds above the
    limit (every candidate fails check_paths -> brake) and an obstacle parked on the path."""
    import torch
    from dcarl_amd import frenet as fr
    g = np.load(GOLD_G)
    fs = fr.FrenetSampler()
    path = fr.ReferencePath(g["wx"], g["wy"], fs.device)
    csp = fo.Spline2D(g["wx"], g["wy"])
    s_end = path.s[-1]
    starts = np.array([[s_end - 20.0, 8.0, 0.5, 0.0, 0.0], [s_end - 2.0, 8.0, 0.0, 0.0, 0.0], [s_end + 5.0, 8.0, 0.0, 0.0, 0.0],
                       [10.0, 30.0, 0.0, 0.0, 0.0], [10.0, 6.0, 0.0, 0.0, 0.0], [40.0, 5.0, -1.0, 0.2, 0.0]])
    cands = fs.calc_frenet_paths(torch.from_numpy(starts), None, None, None, None)
    gp = fr.calc_global_paths(fs, cands, path)
    plen = gp.path_len.cpu().numpy()
    assert plen[2].max() == 0 and 0 < plen[1].max() < 14 and plen[4].min() == 14
    ex, ey = csp.sx.calc(25.0), csp.sy.calc(25.0)
    vehicles = np.array([[ex, ey, 0.0, 0.0, 0.3]])                    # parked on the reference path ahead of start 4 / 3
    for obs in (None, vehicles):
        obs_b = None if obs is None else np.broadcast_to(obs, (len(starts),) + obs.shape).copy()
        choice = fr.get_optimal_trajectory(fs, cands, gp, obs_b).cpu().tolist()
        traj, cost = cands.traj.cpu().numpy(), cands.cost.cpu().numpy()
        ref = [fo.get_optimal_trajectory(traj[b], cost[b], csp, [] if obs is None else obs) for b in range(len(starts))]
        assert choice == ref
    assert ref[3] == 0                                               # 30 m/s start: MAX_SPEED exceeded on every candidate
