"""SURVEY 8(f) rank 4 — episode-return reduction.  The n-step back-up half (RLS.add_data) is PINNED on the reference's own
output file tools/DCARL/visited_value.txt (tests/golden/rls_visited_value.npz: 209 600 rows, 63 complete and 1 622 truncated
gamma runs); the step-reward half (TestScenario_Town03.py, needs `carla`) is restated and checked on hand-checkable episodes
only (oracle/episode_oracle.py header).  GPU: the HIP kernels against the restatement AND against the golden column."""
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


# ---- the back-up pinned on the reference's own output ---------------------------------------------------------------------
def visited_value_golden():
    g = np.load(os.path.join(REPO, "tests", "golden", "rls_visited_value.npz"))
    off = g["ep_off"]
    rew = np.zeros(len(g["value"]))
    rew[off[1:] - 1] = g["terminal_reward"]                  # pre-terminal rewards are 0; the last transition carries -1 or 0
    return g, off, rew


def test_rls_value_stream_reproduces_the_references_visited_value_file():
    """RLS.add_data restated (RlsValueStream) and FED the episodes the file implies must give back the file: all 209 600 rows,
    value column to the file's %f precision (<= 5e-7), action column exactly, in order.  Pins gamma = 0.95 (RLS.py:31), the
    orientation of the exponent (RLS.py:207: the LAST transition gets gamma^0) and the 10-deep buffer (RLS.py:188)."""
    g, off, rew = visited_value_golden()
    assert len(g["value"]) == 209600 and float(g["gamma"]) == 0.95 and int(g["depth"]) == 10
    assert int((g["run_len"] == 10).sum()) == 63 and int(((g["run_len"] > 0) & (g["run_len"] < 10)).sum()) == 1622
    s = eo.RlsValueStream(gamma=float(g["gamma"]))
    act = g["action"]
    for e in range(len(off) - 1):
        for i in range(off[e], off[e + 1]):
            s.add_data(int(i), int(act[i]), float(rew[i]), i == off[e + 1] - 1)
    assert len(s.trajectory_buffer) == 0 and len(s.rows) == 209600
    assert [r[0] for r in s.rows] == list(range(209600))                      # every transition recorded once, in arrival order
    assert np.array_equal(np.array([r[1] for r in s.rows], np.uint8), act)    # [action_to_record, r_to_record] rows
    got = np.array([r[2] for r in s.rows])
    assert np.abs(got - g["value"]).max() <= 5e-7
    # the eleven distinct values of the file are exactly 0 and -0.95^k, k = 0..9, printed with %f
    want = sorted({float("%f" % (-(0.95 ** k))) for k in range(10)} | {0.0})
    assert sorted(np.unique(g["value"]).tolist()) == want
    assert sorted({float("%f" % v) + 0.0 for v in np.unique(got)}) == want
    # a buffer one deeper or shallower, or the exponent the other way round, does NOT reproduce the file
    for depth, flip in ((9, False), (11, False), (10, True)):
        bad = 0
        for e in np.flatnonzero(g["run_len"] == 10)[:5]:
            L = off[e + 1] - off[e]
            k = np.arange(L)[::-1]                                            # distance from the episode's end
            v = np.where(k < depth, -(0.95 ** (depth - 1 - k if flip else k)), 0.0)
            bad += np.abs(v - g["value"][off[e]:off[e + 1]]).max() > 5e-7
        assert bad == 5, (depth, flip)


@pytest.mark.gpu
def test_nstep_backup_kernel_reproduces_the_references_visited_value_file():
    """The HIP kernel on the same 1 793 episodes against the golden column itself (<= 5e-7: the file's %f precision) and against
    the restatement (bit for bit); every transition of a finished episode is recorded."""
    import dcarl_amd as dc
    g, off, rew = visited_value_golden()
    val, rec = dc.episodes.nstep_backup(rew, off, np.ones(len(off) - 1, np.uint8), gamma=float(g["gamma"]), horizon=int(g["depth"]))
    val = val.cpu().numpy()
    assert bool(rec.all())
    assert np.abs(val - g["value"]).max() <= 5e-7
    full = np.flatnonzero(g["run_len"] == 10)
    e = full[0]
    assert val[off[e + 1] - 10:off[e + 1]].tolist() == [-1.0 * 0.95 ** k for k in range(9, -1, -1)]     # == RLS.py:207, bit for bit
    assert float("%f" % val[off[e + 1] - 10]) == -0.630249


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
