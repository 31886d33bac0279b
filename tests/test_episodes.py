"""SURVEY 8(f) rank 4 — episode-return reduction.  CPU: the restatement on hand-checkable episodes (the simulator-dependent
inputs make reference parity unpinned, oracle/episode_oracle.py header).  GPU: the HIP kernels against that restatement."""
import math
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import episode_oracle as eo      # noqa: E402


# ---- the restatement itself, by hand ---------------------------------------------------------------------------------
def test_step_reward_by_hand():
    assert eo.step_reward(3.0, 4.0, False, False, False) == (math.sqrt(5.0) * 0.1, False, 5.0)     # v = 5: 0.1*sqrt(5)
    assert eo.step_reward(0.0, 0.0, False, False, False) == (0.0, False, 0.0)
    assert eo.step_reward(3.0, 4.0, True, False, False)[:2] == (-100, True)                         # collision
    assert eo.step_reward(3.0, 4.0, False, True, False)[:2] == (math.sqrt(5.0) * 0.1, True)         # passed: keeps the step reward
    assert eo.step_reward(3.0, 4.0, False, False, True)[:2] == (0.0, True)                          # stuck
    assert eo.step_reward(3.0, 4.0, False, True, True)[:2] == (math.sqrt(5.0) * 0.1, True)          # `elif`: pass wins over stuck
    assert eo.step_reward(3.0, 4.0, True, False, True)[:2] == (0.0, True)                           # stuck overwrites the -100
    assert eo.step_reward(3.0, 4.0, True, True, False)[:2] == (-100, True)


def test_episode_reward_by_hand():
    # 3 free steps at v = 4 (reward 0.2 each) and a collision: 0.6 - 100; the logged per-action returns of the reference's
    # example file are of this size (a2: -79.05 = a collision after ~210 reward of driving is impossible; ~20.95 of driving)
    tot, ave, rs = eo.episode_reward([(4.0, 0.0, 0, 0, 0)] * 3 + [(0.0, 4.0, 1, 0, 0)])
    assert rs == [0.2, 0.2, 0.2, -100] and tot == pytest.approx(-99.4, abs=1e-12) and ave == 4.0
    assert eo.episode_reward([]) == (0, 0.0, [])


def test_rls_value_stream_by_hand():
    # 13 transitions, rewards 1..13, done at the last: transitions 0,1,2 leave the 10-deep buffer with their OWN reward
    # when transitions 10,11,12 arrive; the remaining ten get 13 * 0.95**k, k = 9 ... 0
    s = eo.RlsValueStream()
    for t in range(13):
        s.add_data(t, t % 3, float(t + 1), t == 12)
    assert [r[0] for r in s.rows] == list(range(13)) and [r[1] for r in s.rows] == [t % 3 for t in range(13)]
    assert [r[2] for r in s.rows[:3]] == [1.0, 2.0, 3.0]
    assert [r[2] for r in s.rows[3:]] == [13.0 * 0.95 ** k for k in range(9, -1, -1)]
    assert s.rows[-1][2] == 13.0 and len(s.trajectory_buffer) == 0
    # a 4-step episode: everything is still buffered at the end
    s = eo.RlsValueStream()
    for t in range(4):
        s.add_data(t, 0, -1.0 if t == 3 else 0.0, t == 3)
    assert [r[2] for r in s.rows] == [-1.0 * 0.95 ** 3, -1.0 * 0.95 ** 2, -1.0 * 0.95, -1.0]
    # no `done`: the last ten stay in the buffer, unrecorded
    s = eo.RlsValueStream()
    for t in range(25):
        s.add_data(t, 0, float(t), False)
    assert [r[0] for r in s.rows] == list(range(15)) and [r[2] for r in s.rows] == [float(t) for t in range(15)]
    assert len(s.trajectory_buffer) == 10


# ---- the HIP kernels ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def dc():
    import dcarl_amd
    dcarl_amd.require_gpu()
    return dcarl_amd


def random_episodes(rng, E, maxlen):
    lens = rng.randint(0, maxlen + 1, E)
    if E > 3:
        lens[rng.randint(0, E, 3)] = 0
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    return lens, off


@pytest.mark.gpu
@pytest.mark.parametrize("E,maxlen,seed", [(1, 5, 0), (40, 300, 1), (1000, 70, 2), (3, 5000, 3)])
def test_episode_returns_vs_restatement(dc, E, maxlen, seed):
    rng = np.random.RandomState(seed)
    lens, off = random_episodes(rng, E, maxlen)
    N = int(off[-1])
    vx, vy = rng.uniform(-12, 12, N), rng.uniform(-12, 12, N)
    if N:
        vx[rng.randint(0, N, 5)] = 0.0
        vy[rng.randint(0, N, 5)] = 0.0
    flags = np.zeros(N, np.uint8)
    for e in range(E):                                      # the terminal step of some episodes carries the done cause
        if lens[e]:
            flags[off[e + 1] - 1] = rng.choice([0, 1, 2, 4, 5, 6, 3, 7])
    ep, sp, st = dc.episodes.episode_returns(vx, vy, flags, off)
    ep, sp, st = ep.cpu().numpy(), sp.cpu().numpy(), st.cpu().numpy()
    for e in range(E):
        steps = [(vx[i], vy[i], flags[i] & 1, flags[i] & 2, flags[i] & 4) for i in range(off[e], off[e + 1])]
        tot, ave, rs = eo.episode_reward(steps)
        got, want = st[off[e]:off[e + 1]], np.array(rs, dtype=np.float64)
        # two nested IEEE square roots: the device's are within 1 ulp of libm's (identical on almost every input)
        ulp = np.abs(got - want) / np.maximum(np.spacing(np.abs(want)), 1e-300)
        assert ulp.max(initial=0.0) <= 2.0, (ulp.max(), int((ulp > 0).sum()), len(ulp))
        assert np.array_equal(got[want <= 0.0], want[want <= 0.0])                        # -100 / 0.0 are exact
        assert abs(ep[e] - tot) <= 1e-12 * max(1.0, abs(tot))                             # tree sum vs running sum
        assert abs(sp[e] - ave) <= 1e-12 * max(1.0, abs(ave))
    ep2, _, none = dc.episodes.episode_returns(vx, vy, flags, off, want_steps=False)
    assert none is None and np.array_equal(ep2.cpu().numpy(), ep)                         # run-to-run identical


@pytest.mark.gpu
@pytest.mark.parametrize("E,maxlen,seed", [(1, 13, 0), (60, 40, 1), (500, 9, 2), (4, 3000, 3)])
def test_nstep_backup_bit_exact_vs_restatement(dc, E, maxlen, seed):
    rng = np.random.RandomState(seed)
    lens, off = random_episodes(rng, E, maxlen)
    N = int(off[-1])
    rew = np.where(rng.rand(N) < 0.7, 0.0, rng.uniform(-1, 0, N))
    done = (rng.rand(E) < 0.8).astype(np.uint8)
    val, rec = dc.episodes.nstep_backup(rew, off, done)
    val, rec = val.cpu().numpy(), rec.cpu().numpy()
    for e in range(E):
        s = eo.RlsValueStream()
        for i in range(off[e], off[e + 1]):
            s.add_data(int(i), 0, float(rew[i]), bool(done[e]) and i == off[e + 1] - 1)
        ids = [r[0] for r in s.rows]
        assert ids == [int(i) for i in range(off[e], off[e + 1]) if rec[i]]               # which transitions get recorded
        assert [r[2] for r in s.rows] == val[ids].tolist()                                # and with which value: bit-exact
    assert np.array_equal(dc.episodes.gamma_powers(0.95, 10), np.array([0.95 ** k for k in range(10)]))
    # other gamma / horizon
    val2, rec2 = dc.episodes.nstep_backup(rew, off, np.ones(E, np.uint8), gamma=0.9, horizon=3)
    for e in range(E):
        n = lens[e]
        for k in range(min(3, n)):
            assert val2[off[e + 1] - 1 - k].item() == rew[off[e + 1] - 1] * 0.9 ** k
    assert bool(rec2.all())


@pytest.mark.gpu
def test_episode_api_rejects_bad_offsets(dc):
    with pytest.raises(ValueError):
        dc.episodes.episode_returns([1.0, 2.0], [0.0, 0.0], [0, 0], [0, 3])
    with pytest.raises(ValueError):
        dc.episodes.nstep_backup([1.0, 2.0], [0, 2], [1, 1])
    ep, sp, st = dc.episodes.episode_returns([], [], [], [0])
    assert ep.numel() == 0 and st.numel() == 0
