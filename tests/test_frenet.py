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
