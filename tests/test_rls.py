"""Field variant of the confidence test: oracle vs the field-log goldens (CPU) and the HIP
scan / decision kernels vs the oracle (GPU)."""
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import rls_oracle as ro  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden", "rls_field_decisions.npz")


def test_oracle_decision_reproduces_the_field_log():
    """70 records of Field_testing/Scenario{2,3}/RLS.txt in which the vehicle executed an RL action: with the logged
    statistics the restated act_test must choose the RL candidate (and must refuse once the rule action looks good,
    has too few visits, or the candidate has too few)."""
    g = np.load(GOLD)
    assert len(g["action"]) == 70 and np.all(g["action"] == 1)
    for i in range(70):
        count = [g["n_rule"][i], g["n_rl"][i]]
        mean = [g["mean_rule"][i], g["mean_rl"][i]]
        var = [g["var_rule"][i], g["var_rl"][i]]
        assert ro.act_test_from_stats(count, mean, var) == 1
        assert ro.act_test_from_stats([29, count[1]], mean, var) == 0            # RLS:141 visited_times_rule < 30
        assert ro.act_test_from_stats([count[0], 4], mean, var) == 0             # RLS:141 visited_times_RL < 5
        assert ro.act_test_from_stats(count, [-0.05, mean[1]], var) == 0         # RLS:141 mean_rule > -0.1
        assert ro.act_test_from_stats(count, [mean[1] + 0.01, mean[1]], var) == 0  # candidate not better: cdf < 0.5


def test_oracle_refuses_where_the_field_log_shows_the_rule_action():
    """The NEGATIVE decisions of the same logs: 3 732 records (660 distinct statistics) in which the vehicle executed the
    rule action while one of the two gates of RLS:141 that look at the rule action alone was closed — plus (round 4) the 107
    rows of tools/DCARL/driving_record.txt, the reference's own log in the RLS.py:217-241 format: all rule actions, all with a
    closed gate (80 more distinct statistics).  Whatever the
    candidates' statistics are — here the most favourable ones imaginable — act_test must return 0."""
    g = np.load(GOLD)
    n, m, v = g["neg_n_rule"], g["neg_mean_rule"], g["neg_var_rule"]
    assert len(n) == 740 and int(g["distinct_from_field_logs"]) == 660 and int(g["rule_rows_with_open_gates"]) == 5
    assert int(g["driving_record_rows"]) == 107 == int(g["driving_record_closed"]) and int(g["driving_record_open"]) == 0
    assert np.all((n < 30) | (m > -0.1))
    rng = np.random.RandomState(0)
    for i in range(len(n)):
        count = [n[i]] + [10 ** 6] * 7                       # every candidate visited a million times,
        mean = [m[i]] + [0.0] * 7                            # with the best possible value
        var = [max(v[i], 0.0)] + list(rng.rand(7) * 1e-6)    # and almost no spread
        if n[i] == 0:
            mean[0], var[0] = -1.0, -1.0                     # RLS:167-168
        assert ro.act_test_from_stats(count, mean, var) == 0


def test_oracle_neighbour_stats_small_case():
    st = np.zeros((3, 21)); st[1, 0] = 0.9; st[2, 0] = 5.0; st[:, 20] = [0, 0, 1]
    val = np.array([-0.2, -0.4, -0.9])
    q = np.zeros((3, 21)); q[1, 20] = 1.0; q[1, 0] = 5.5; q[2, 0] = 100.0
    n, m, v = ro.neighbour_stats(st, val, q)
    assert n.tolist() == [2, 1, 0]
    assert abs(m[0] + 0.3) < 1e-15 and abs(v[0] - 0.01) < 1e-15 and m[1] == -0.9 and v[1] == 0.0
    assert m[2] == -1.0 and v[2] == -1.0                                          # RLS:167-168


def _synthetic_table(rng, N):
    """Rows shaped like the field data: four surrounding vehicles that are either absent (the constants the log shows)
    or jittered around a few prototypes, actions 0..7, values in (-1, 0)."""
    proto = rng.uniform(-20, 20, (12, 20))
    st = proto[rng.randint(0, 12, N)] + rng.normal(0, 0.4, (N, 20)) * np.array(ro.VISITED_STATE_DIST[:20])
    absent = rng.rand(N, 4) < 0.3
    for v in range(4):
        st[absent[:, v], 4 + 4 * v: 8 + 4 * v] = [50.0, 1.0, 20.0, 0.0]
    act = rng.randint(0, 8, N).astype(np.float64)
    states = np.round(np.column_stack([st, act]), 6)            # visited_state.txt is written with %f
    values = np.round(-rng.rand(N), 6)
    return states, values


@pytest.mark.gpu
@pytest.mark.parametrize("N,Q,seed", [(1, 1, 0), (5000, 300, 1), (2049, 257, 2), (20000, 64, 3)])
def test_neighbour_stats_vs_oracle(N, Q, seed):
    import dcarl_amd as dc
    rng = np.random.RandomState(seed)
    states, values = _synthetic_table(rng, N)
    pick = rng.randint(0, N, Q)
    queries = states[pick].copy()
    if N > 1:
        queries[:, :20] += rng.normal(0, 0.5, (Q, 20)) * np.array(ro.VISITED_STATE_DIST[:20])
    queries[: Q // 8] = states[pick[: Q // 8]] + np.array(ro.VISITED_STATE_DIST) * rng.choice([-1.0, 1.0], (Q // 8, 21))  # on the box faces
    rls = dc.rls.RLS(states, np.column_stack([states[:, 20], values]))
    n, m, v = rls.statistics(queries)
    rn, rm, rv = ro.neighbour_stats(states, values, queries)
    assert np.array_equal(n.cpu().numpy(), rn)                  # bit-exact membership, including the closed faces
    assert rn.max() > (1 if N > 1000 else 0)
    assert np.allclose(m.cpu().numpy(), rm, rtol=0, atol=1e-13) and np.allclose(v.cpu().numpy(), rv, rtol=0, atol=1e-13)
    n2, m2, v2 = rls.statistics(queries)                        # fixed-order reduction: run-to-run identical
    assert np.array_equal(m2.cpu().numpy(), m.cpu().numpy()) and np.array_equal(v2.cpu().numpy(), v.cpu().numpy())


@pytest.mark.gpu
def test_decide_vs_field_log_and_oracle():
    import torch
    import dcarl_amd as dc
    g = np.load(GOLD)
    rls = dc.rls.RLS(np.zeros((1, 21)), np.zeros(1))
    dev = rls.device
    # the 70 logged decisions (one candidate each) ...
    count = torch.tensor(np.column_stack([g["n_rule"], g["n_rl"]]).astype(np.int64), device=dev)
    mean = torch.tensor(np.column_stack([g["mean_rule"], g["mean_rl"]]), device=dev)
    var = torch.tensor(np.column_stack([g["var_rule"], g["var_rl"]]), device=dev)
    assert rls.decide(count, mean, var, 1).cpu().tolist() == [1] * 70
    # ... the 660 distinct logged NEGATIVE decisions (rule action executed, a rule-side gate closed), each against seven
    # candidates with the most favourable statistics imaginable ...
    nn, nm, nv = g["neg_n_rule"], g["neg_mean_rule"], g["neg_var_rule"]
    cnt = np.column_stack([nn] + [np.full(len(nn), 10 ** 6)] * 7).astype(np.int64)
    mu = np.column_stack([np.where(nn == 0, -1.0, nm)] + [np.zeros(len(nn))] * 7)
    va = np.column_stack([np.where(nn == 0, -1.0, np.maximum(nv, 0.0))] + [np.full(len(nn), 1e-7)] * 7)
    neg = rls.decide(torch.tensor(cnt, device=dev), torch.tensor(mu, device=dev), torch.tensor(va, device=dev), 7)
    assert neg.cpu().tolist() == [0] * len(nn)
    # ... and random statistics with 7 candidates against the restated act_test, degenerate cases included
    rng = np.random.RandomState(5)
    B = 4000
    cnt = rng.randint(0, 80, (B, 8)).astype(np.int64)
    mu = -rng.rand(B, 8)
    va = rng.rand(B, 8) * 0.3
    va[:200] = 0.0                                              # sd == 0: +-inf / NaN z
    mu[:100, 1:] = mu[:100, :1]                                 # ... with equal means (0/0)
    mu[cnt == 0] = -1.0; va[cnt == 0] = -1.0
    ref = [ro.act_test_from_stats(cnt[b], mu[b], va[b]) for b in range(B)]
    got = rls.decide(torch.tensor(cnt, device=dev), torch.tensor(mu, device=dev), torch.tensor(va, device=dev), 7)
    assert got.cpu().tolist() == ref
    assert len(set(ref)) > 3


@pytest.mark.gpu
def test_act_test_end_to_end_vs_oracle():
    import dcarl_amd as dc
    rng = np.random.RandomState(9)
    states, values = _synthetic_table(rng, 30000)
    values[states[:, 20] == 0] -= 0.3                           # make the rule action look bad so that candidates win
    rls = dc.rls.RLS(states, values, visited_times_thres=5)
    obs = states[rng.randint(0, len(states), 200), :20]
    got = rls.act_test(obs).cpu().tolist()
    ref = []
    for o in obs:
        q = np.stack([np.append(o, a) for a in range(8)])
        n, m, v = ro.neighbour_stats(states, values, q)
        ref.append(ro.act_test_from_stats(n, m, v, visited_times_thres=5))
    assert got == ref and len(set(ref)) > 1


@pytest.mark.gpu
def test_empty_table_and_no_queries():
    import dcarl_amd as dc
    rls = dc.rls.RLS(np.zeros((0, 21)), np.zeros(0))
    n, m, v = rls.statistics(np.zeros((3, 21)))
    assert n.cpu().tolist() == [0, 0, 0] and m.cpu().tolist() == [-1.0] * 3 and v.cpu().tolist() == [-1.0] * 3   # RLS:167-168
    assert rls.act_test(np.zeros((2, 20))).cpu().tolist() == [0, 0]
    n, m, v = rls.statistics(np.zeros((0, 21)))
    assert n.numel() == 0


@pytest.mark.gpu
def test_train_time_gate_vs_oracle_and_seeded_draws():
    """act / act_train / should_use_rule (RLS:78-118) on the GPU statistics: against the restatement with injected exploration
    draws, and with Python's random seeded like one would seed the reference (one uniform(-1, 0) per observation that passed
    the visit-count test, in order)."""
    import random
    import torch
    from dcarl_amd import rls as drls
    rng = np.random.RandomState(3)
    N, B = 4000, 300
    proto = rng.uniform(-5, 5, (6, 20))
    dist = np.array(drls.VISITED_STATE_DIST)
    st = proto[rng.randint(0, 6, N)] + rng.normal(0, 0.3, (N, 20)) * dist[:20]
    states = np.column_stack([st, rng.randint(0, 3, N).astype(np.float64)])
    values = -rng.rand(N)
    r = drls.RLS(states, values, visited_times_thres=20, is_training=True)
    obs = np.concatenate([proto[rng.randint(0, 6, B - 20)] + rng.normal(0, 0.2, (B - 20, 20)) * dist[:20],
                          rng.uniform(50, 60, (20, 20))])                      # the last 20: never-visited states
    cnt, mean = (t.cpu().numpy() for t in r._rule_statistics(obs))
    assert (cnt[-20:] == 0).all() and (cnt[:-20] >= 20).any()
    explore = rng.uniform(-1, 0, B)
    rl = rng.randint(1, 8, B)
    want_use = [ro.should_use_rule_from_stats(cnt[b], mean[b], explore[b], 20) for b in range(B)]
    assert r.should_use_rule(obs, explore).cpu().tolist() == want_use
    assert 0.1 < np.mean(want_use) < 0.9
    want_act = [ro.act_train_from_stats(cnt[b], mean[b], explore[b], rl[b], 20) for b in range(B)]
    assert r.act_train(obs, rl, explore).cpu().tolist() == want_act
    assert r.act(obs, rl, explore).cpu().tolist() == want_act                  # is_training -> act_train
    # seeded like the reference: random.uniform(-1, 0) once per observation with enough visits, in order
    random.seed(11)
    got = r.act_train(obs, rl).cpu().tolist()
    random.seed(11)
    ref = []
    for b in range(B):
        if cnt[b] < 20:
            ref.append(0)
            continue
        ref.append(0 if random.uniform(-1, 0) < mean[b] else int(rl[b]))
    assert got == ref
    r.is_training = False
    assert r.act(obs, rl).cpu().tolist() == r.act_test(obs).cpu().tolist()   # RLS:81-82
