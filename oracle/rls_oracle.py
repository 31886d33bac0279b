"""CPU restatement (NumPy, float64) of the reference's field confidence test -- TEST INFRASTRUCTURE ONLY.

Follows Field_testing/Software_and_Raw_Data_on_Self-Driving_Vehicle/software/src/tools/DCARL/stable_baselines/deepq/RLS.py:
  box insert RLS:193-194, visited times RLS:161-163, statistics RLS:165-181, act_test RLS:120-157.

PARITY: the neighbour search is UNPINNED -- the reference delegates it to the third-party `rtree` package
(libspatialindex; imported at RLS:5, pinned in no requirements file, not installed here) and the visited-state table it
was run on (visited_state.txt / state_index.*) is absent from the reference checkout (.MISSING_LARGE_BLOBS).  The
restatement below is the documented semantics of `Index.intersection` for a point query against closed boxes.
The DECISION arithmetic is pinned: tests/golden/rls_field_decisions.npz holds the 70 field-log records of
Field_testing/Scenario{2,3}/RLS.txt in which the vehicle executed an RL action together with the statistics the
reference logged for them (RLS:224-241), and `act_test_from_stats` must reproduce "take the RL action" on all of them."""
from __future__ import annotations

import math

import numpy as np

VISITED_STATE_DIST = np.array([1, 0.3, 2, 50, 10, 0.3, 2, 50, 10, 0.3, 2, 50, 10, 0.3, 2, 50, 10, 0.3, 2, 50, 0.1])  # RLS:68


def neighbour_stats(states, values, queries, half=VISITED_STATE_DIST):
    """count / mean / var per query point; (-1, -1) where nothing is visited (RLS:165-168)."""
    states = np.asarray(states, np.float64).reshape(-1, 21)
    values = np.asarray(values, np.float64)
    queries = np.asarray(queries, np.float64).reshape(-1, 21)
    lo, hi = states - half, states + half                     # RLS:193-194: the box that was inserted
    count = np.zeros(len(queries), np.int64)
    mean = np.full(len(queries), -1.0)
    var = np.full(len(queries), -1.0)
    for i, q in enumerate(queries):
        hit = np.all((lo <= q) & (q <= hi), axis=1)            # point in closed box (rtree Index.intersection)
        n = int(hit.sum())
        count[i] = n
        if n:
            v = values[hit]
            mean[i], var[i] = np.mean(v), np.var(v)            # RLS:175-176
    return count, mean, var


def norm_cdf(z):
    return 0.5 * math.erfc(-z / math.sqrt(2.0))               # scipy.stats.norm.cdf


def act_test_from_stats(count, mean, var, visited_times_thres=30, min_rl_visits=5, rule_mean_gate=-0.1,
                        confidence_thres=0.5):
    """RLS:120-157 for one decision: index 0 = rule action, index c = candidate c."""
    n_rule, mean_rule, var_rule = count[0], mean[0], var[0]
    for c in range(1, len(count)):
        if n_rule < visited_times_thres or count[c] < min_rl_visits or mean_rule > rule_mean_gate:   # RLS:141
            continue
        var_diff = var_rule / n_rule + var[c] / count[c]       # RLS:144
        with np.errstate(all="ignore"):
            z = (mean[c] - mean_rule) / np.sqrt(var_diff)      # RLS:145-148
        if norm_cdf(z) > confidence_thres:                     # RLS:150
            return c
    return 0


def should_use_rule_from_stats(count_rule, mean_rule, explore_motivation, visited_times_thres=30):
    """RLS:94-116 for one observation, the random.uniform(-1, 0) draw injected (the reference draws it only when the first
    test did not already return, RLS:107-112)."""
    if count_rule < visited_times_thres:                       # RLS:107-108
        return True
    if explore_motivation < mean_rule:                         # RLS:112-114
        return True
    return False


def act_train_from_stats(count_rule, mean_rule, explore_motivation, rl_action, visited_times_thres=30):
    """RLS:85-89."""
    return 0 if should_use_rule_from_stats(count_rule, mean_rule, explore_motivation, visited_times_thres) else int(rl_action)
