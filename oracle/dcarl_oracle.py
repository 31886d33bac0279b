"""CPU oracle for the DCARL confidence-estimation hot path (TEST INFRASTRUCTURE).

A NumPy float64 restatement of the reference algorithm.  Every function cites
the reference lines it follows; abbreviations (relative to the reference root):

    S1 = Simulation_testing/Simulation_1/test_DCARL.py
    S2 = Simulation_testing/Simulation_2/test_DCARL.py
    DS = Simulation_testing/Simulation_Data_Collection/Data_Sampling/data_sampling.py

Parity status: PINNED.  ``tests/golden/make_goldens.py`` runs the unmodified
reference scripts/functions in the build container and commits their outputs
as fixtures under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks
every function here against them (bit-exact for the structure-faithful forms,
<=1e-11 abs for the O(1)-per-record forms, arg-max exact).

Two formulations of the online loop are provided:

* ``run_online_faithful`` keeps the reference's structure (append to a Python
  list, re-materialise the bucket and recompute mean/std from scratch for every
  record) — bit-exact with the reference, O(N^2/(S*A)), used for goldens and as
  the "same algorithmic structure" CPU timing.
* ``run_online_sums`` keeps (n, sum, sum of squares) per bucket — the
  formulation the HIP kernels implement — O(1) per record.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np


# --------------------------------------------------------------------------
# parameters (reference: hard-coded literals, S1:10 defaults, S1:43-52)
# --------------------------------------------------------------------------
@dataclass(frozen=True)
class Params:
    rule_act: int = 0        # S1:43  "rule_act = 0"
    n_thres: int = 10        # S1:45  update only when len(bucket) > n_thres
    alpha: float = 0.05      # S1:10  default arg
    scale: float = 150.0     # S1:10  default arg
    cap: float = 100.0       # S1:12  min(100, ...)
    init_rule: float = 100.0  # S1:52 temp[rule_act] = 100
    init_other: float = -50.0  # S1:51 temp = [-50]*action_num


DEFAULT = Params()


def hoeffding_halfwidth(n, alpha=0.05, scale=150.0):
    """scale*sqrt(log(1/alpha)/2/n), same operation order as S1:12/S1:16."""
    return scale * math.sqrt(math.log(1 / alpha) / 2 / n)


# --------------------------------------------------------------------------
# a1-a4: the four bound functions (S1:10-28 == S2:9-27)
# --------------------------------------------------------------------------
def upper_bound(x, alpha=0.05, loc=-50, scale=150):
    """S1:10-12.  Hoeffding upper bound on the mean, capped at 100. ``loc`` unused."""
    x = np.asarray(x, dtype=np.float64)
    return min(100, np.mean(x) + hoeffding_halfwidth(len(x), alpha, scale))


def lower_bound(x, alpha=0.05, loc=-50, scale=150):
    """S1:14-16.  Hoeffding lower bound on the mean (no floor)."""
    x = np.asarray(x, dtype=np.float64)
    return np.mean(x) - hoeffding_halfwidth(len(x), alpha, scale)


def CI_lower_bound(x, alpha=0.05, loc=-50, scale=150):
    """S1:18-24.  n+1 "pseudo-sample" lower bound with a 4-sigma/(n+1) penalty.

    sigma is the population standard deviation (np.std, ddof=0).
    """
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    total = np.sum(x)
    sigma = np.std(x)
    return (total / n / (n + 1) - 4 * sigma / (n + 1) + total / (n + 1)
            - hoeffding_halfwidth(n + 1, alpha, scale))


def mean_value(x, alpha=0.05, loc=-50, scale=150):
    """S1:26-28.  min(100, mean) — defined by the reference, never called."""
    x = np.asarray(x, dtype=np.float64)
    return min(100, np.mean(x))


def bucket_value(x, is_rule, p: Params = DEFAULT):
    """S1:87-90: rule action gets the UCB, every other candidate min(LB, CI-LB)."""
    if is_rule:
        return min(p.cap, np.mean(x) + hoeffding_halfwidth(len(x), p.alpha, p.scale))
    return min(lower_bound(x, p.alpha, scale=p.scale), CI_lower_bound(x, p.alpha, scale=p.scale))


# --------------------------------------------------------------------------
# closed form from sufficient statistics (what the HIP kernels evaluate)
# --------------------------------------------------------------------------
def value_from_sums(n, s, q, is_rule, p: Params = DEFAULT):
    """V from (n, sum, sum of squares) in float64.

    Algebraically identical to S1:10-24; the population variance is
    q/n - mean^2 (clamped at 0) instead of NumPy's two-pass form.
    """
    mean = s / n
    hw = hoeffding_halfwidth(n, p.alpha, p.scale)
    if is_rule:
        return min(p.cap, mean + hw)
    var = max(q / n - mean * mean, 0.0)
    sigma = math.sqrt(var)
    lb = mean - hw
    ci = (s / n / (n + 1) - 4 * sigma / (n + 1) + s / (n + 1)
          - hoeffding_halfwidth(n + 1, p.alpha, p.scale))
    return min(lb, ci)


def bounds_batch(values, seg_off, S, A, p: Params = DEFAULT):
    """Final-state ("batch") evaluation: samples sorted by (state, action).

    values[seg_off[s*A+a]:seg_off[s*A+a+1]] is bucket (s,a).  Returns
    V (S,A) f64, n (S,A) i64, vmax (S,), amax (S,) with the reference's
    first-max tie rule (S1:93-94) and the threshold rule of S1:86.
    """
    values = np.asarray(values, dtype=np.float64)
    V = np.full((S, A), p.init_other, dtype=np.float64)
    V[:, p.rule_act] = p.init_rule
    cnt = np.zeros((S, A), dtype=np.int64)
    for s in range(S):
        for a in range(A):
            b, e = int(seg_off[s * A + a]), int(seg_off[s * A + a + 1])
            cnt[s, a] = e - b
            if e - b > p.n_thres:
                V[s, a] = bucket_value(values[b:e], a == p.rule_act, p)
    amax = np.argmax(V, axis=1)
    vmax = V[np.arange(S), amax]
    return V, cnt, vmax, amax


# --------------------------------------------------------------------------
# a5-a10: the online loop
# --------------------------------------------------------------------------
def _init_tables(S, A, p):
    V = [[p.init_other] * A for _ in range(S)]          # S1:50-53
    for s in range(S):
        V[s][p.rule_act] = p.init_rule
    return V


def run_online_faithful(data, S, A, p: Params = DEFAULT, limit=20000,
                        true_action_values=None, with_overall=False):
    """Structure-faithful restatement of S1:73-99 / S2:72-105.

    ``data`` is the (N,4) float64 record table {state idx, state feature,
    action, cumulative reward} (a11); only ``data[:limit]`` is consumed
    (S1:73).  Returns a dict with the reference's script-level globals.
    """
    buckets = [[[] for _ in range(A)] for _ in range(S)]     # S1:41
    V = _init_tables(S, A, p)
    step_value = [[] for _ in range(S)]                      # S1:47
    step_act = [[] for _ in range(S)]                        # S1:57
    true_step_value = [[] for _ in range(S)]                 # S1:56
    activation_step = np.full(S, -1, dtype=np.int64)         # S1:58
    activation_value = np.full(S, -1, dtype=np.int64)        # S1:59 (never written)
    overall = []
    for idx_f, _feat, act_f, r in data[0:limit]:             # S1:73
        s, a = int(idx_f), int(act_f)                        # S1:77-78
        b = buckets[s][a]
        b.append(r)                                          # S1:80
        if len(b) > p.n_thres:                               # S1:86
            x = np.array(b)
            if a == p.rule_act:                              # S1:87-88
                V[s][a] = min(p.cap, np.mean(x) + hoeffding_halfwidth(len(x), p.alpha, p.scale))
            else:                                            # S1:89-90
                V[s][a] = min(lower_bound(x, p.alpha, scale=p.scale),
                              CI_lower_bound(x, p.alpha, scale=p.scale))
        row = np.array(V[s])
        step_value[s].append(max(row))                       # S1:93
        best = int(np.argmax(row))                           # S1:94 (first max wins)
        step_act[s].append(best)                             # S1:95
        if true_action_values is not None:
            true_step_value[s].append(true_action_values[s][best])   # S1:96
        if activation_step[s] == -1 and best != p.rule_act:  # S1:98-99
            activation_step[s] = len(step_value[s])
        if with_overall:                                     # S2:99-105
            tot = 0
            for i in range(S):
                if activation_step[i] != -1:
                    tot = tot + max(np.array(V[i])) - activation_value[i] * 0.9
            overall.append(tot)
    return dict(TSRL_value=V, step_TSRL_value=step_value, step_TSRL_act=step_act,
                true_step_TSRL_value=true_step_value, activation_step=activation_step,
                overall_value=overall,
                bucket_len=[[len(buckets[s][a]) for a in range(A)] for s in range(S)])


def run_online_sums(state, act, reward, S, A, p: Params = DEFAULT, with_overall=False):
    """O(1)-per-record formulation (the one the HIP kernels implement).

    state/act/reward are 1-D arrays in arrival order.  Same outputs as
    ``run_online_faithful`` but traces come back as flat arrays in arrival
    order plus per-state views.
    """
    N = len(reward)
    cnt = np.zeros((S, A), dtype=np.int64)
    sm = np.zeros((S, A), dtype=np.float64)
    sq = np.zeros((S, A), dtype=np.float64)
    V = np.full((S, A), p.init_other, dtype=np.float64)
    V[:, p.rule_act] = p.init_rule
    seen = np.zeros(S, dtype=np.int64)
    activation_step = np.full(S, -1, dtype=np.int64)
    step_val = np.empty(N, dtype=np.float64)
    step_act = np.empty(N, dtype=np.int64)
    overall = np.zeros(N, dtype=np.float64)
    cur_max = np.zeros(S, dtype=np.float64)
    for k in range(N):
        s, a, r = int(state[k]), int(act[k]), float(reward[k])
        cnt[s, a] += 1
        sm[s, a] += r
        sq[s, a] += r * r
        n = int(cnt[s, a])
        if n > p.n_thres:
            V[s, a] = value_from_sums(n, sm[s, a], sq[s, a], a == p.rule_act, p)
        best = int(np.argmax(V[s]))
        step_val[k] = V[s, best]
        step_act[k] = best
        seen[s] += 1
        if activation_step[s] == -1 and best != p.rule_act:
            activation_step[s] = seen[s]
        if with_overall:
            cur_max[s] = V[s, best]
            m = activation_step != -1
            overall[k] = float(np.sum(cur_max[m] + 0.9)) if m.any() else 0.0
    return dict(V=V, n=cnt, step_val=step_val, step_act=step_act,
                activation_step=activation_step, overall_value=overall)


def trace_lengths_sorted(lengths):
    """S2:108-119: (state id, trace length) rows sorted by length descending.

    The reference uses ``np.argsort(-len)`` (quicksort, not stable); for the
    bundled data all lengths are distinct so the order is unambiguous.
    """
    lengths = np.asarray(lengths)
    arr = np.stack([np.arange(len(lengths)), lengths], axis=1)
    return arr[np.argsort(-lengths)]


# --------------------------------------------------------------------------
# a12-a15: Monte-Carlo return sampler, restated on explicit noise streams
# --------------------------------------------------------------------------
def add_an_act_data_from_noise(q_row, act, z, sigma=50.0):
    """DS:5-9: norm.rvs(loc=Q[act], scale=50) == Q[act] + 50*z (two roundings)."""
    return float(q_row[act] + sigma * z)


def random_state_norm_from_noise(state_num, z):
    """DS:12-17: floor((3 + 1*z)/6*state_num).astype(int); may be out of range."""
    v = 3.0 + 1.0 * np.asarray(z, dtype=np.float64)
    return np.floor(v / 6 * state_num).astype(int)


def random_state_manual_from_streams(u, r):
    """DS:19-28 with the draws injected: u[i] = the i-th random.random(), r[j] = the j-th random.randint(1, state_num-1)
    (the reference draws one only when u[i] > 0.1, DS:22-23).  Returns the list DS:28 returns."""
    out, j = [], 0
    for x in np.asarray(u, dtype=np.float64):
        if x > 0.1:                                                        # DS:22
            out.append(int(r[j]))                                          # DS:23
            j += 1
        else:
            out.append(0)                                                  # DS:25
    return out


def data_generation_from_streams(u_states, u_q, z_visit, acts, z_reward,
                                 state_num=20, action_num=11, lo=-50.0, hi=100.0, sigma=50.0):
    """DS:30-67 with every random draw injected.

    u_states (state_num,), u_q (state_num, action_num) are U[0,1) draws;
    z_visit (data_size,) standard normals; acts / z_reward one entry per KEPT
    visit.  Returns (data (M,4), action_values (S,A), states (S,)).
    """
    states = 0.0 + 1.0 * np.asarray(u_states, dtype=np.float64)           # DS:39
    q = lo + (hi - lo) * np.asarray(u_q, dtype=np.float64)                # DS:42-43
    idxs = random_state_norm_from_noise(state_num, z_visit)               # DS:45
    rows = []
    j = 0
    for idx in idxs:                                                      # DS:49
        if idx < 0 or idx >= state_num:                                   # DS:50-51
            continue
        a = int(acts[j])                                                  # DS:54
        r = add_an_act_data_from_noise(q[idx], a, z_reward[j], sigma)     # DS:55
        rows.append([int(idx), states[idx], a, r])
        j += 1
    return np.array(rows), q, states


def data_generation_seeded(seed, state_num=20, data_size=50000, action_num=11):
    """Replays the reference's draw ORDER on seeded legacy generators.

    np.random.seed(seed); random.seed(seed); then exactly the calls the
    reference makes (DS:39,43,45,54,55), expressed as RandomState primitives:
    uniform.rvs -> random_sample, norm.rvs -> standard_normal.
    """
    import random as pyrandom
    rs = np.random.RandomState(seed)
    pyrandom.seed(seed)
    u_states = rs.random_sample(state_num)
    u_q = np.stack([rs.random_sample(action_num) for _ in range(state_num)])
    z_visit = rs.standard_normal(data_size)
    idxs = random_state_norm_from_noise(state_num, z_visit)
    acts, zs = [], []
    for idx in idxs:
        if idx < 0 or idx >= state_num:
            continue
        acts.append(pyrandom.randint(0, action_num - 1))
        zs.append(rs.standard_normal(1)[0])
    data, q, states = data_generation_from_streams(u_states, u_q, z_visit, acts, zs,
                                                   state_num, action_num)
    return data, q, states, dict(u_states=u_states, u_q=u_q, z_visit=z_visit,
                                 acts=np.array(acts), z_reward=np.array(zs))


# --------------------------------------------------------------------------
# Philox-4x32-10 + Box-Muller (the build's own counter RNG; Salmon et al. SC'11)
# --------------------------------------------------------------------------
_PH_M0, _PH_M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_PH_W0, _PH_W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox-4x32-10.  All inputs broadcastable uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = c0.astype(np.uint64) * _PH_M0
            p1 = c2.astype(np.uint64) * _PH_M1
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(_PH_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_PH_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def u32_to_unit_open(x):
    """(x + 0.5) * 2^-32 in float64: uniform on (0,1), never 0 or 1."""
    return (np.asarray(x, dtype=np.float64) + 0.5) * (1.0 / 4294967296.0)


def box_muller(x1, x2):
    """Two standard normals from two uint32 words (float64 math)."""
    u1, u2 = u32_to_unit_open(x1), u32_to_unit_open(x2)
    r = np.sqrt(-2.0 * np.log(u1))
    th = 2.0 * np.pi * u2
    return r * np.cos(th), r * np.sin(th)


def mulhi_u32(x, m):
    """floor(x*m / 2^32): maps a uint32 word to {0..m-1}."""
    return ((np.asarray(x, dtype=np.uint64) * np.uint64(m)) >> np.uint64(32)).astype(np.int64)


def sample_state_records(q, T, seed, sigma=50.0, stream=0):
    """The build's per-state record sampler (dcarl_sample_state_records).

    Record t of state s draws Philox(ctr=(t, s, stream, 0), key=(seed lo, hi));
    act = mulhi(x0, A); R = Q[s,act] + sigma*z with z the cosine Box-Muller
    branch of (x1, x2).  Returns act (S,T) int64, R (S,T) float64, z (S,T).
    """
    q = np.asarray(q, dtype=np.float64)
    S, A = q.shape
    t = np.arange(T, dtype=np.uint32)[None, :]
    s = np.arange(S, dtype=np.uint32)[:, None]
    x0, x1, x2, _x3 = philox4x32_10(t, s, np.uint32(stream), np.uint32(0),
                                    seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    act = mulhi_u32(x0, A)
    z, _ = box_muller(x1, x2)
    r = np.take_along_axis(q, act, axis=1) + sigma * z
    return act, r, z


def sample_pairs(q, N, seed, offset=0, sigma=50.0, stream=1, state_loc=3.0, state_scale=1.0,
                 state_div=6.0):
    """The build's {s,a,R} pair sampler (dcarl_sample_pairs), DS:45-55 semantics.

    Draw g = offset+i is draw k = g%4 of group G = g//4; the group owns the 12 words of the Philox blocks with
    64-bit counters 3G, 3G+1, 3G+2 (ctr = (lo, hi, stream, 0)) and draw k uses words 3k (action), 3k+1, 3k+2:
    act = mulhi(w[3k], A); (zR, zS) = Box-Muller(w[3k+1], w[3k+2]);
    idx = floor((loc + scale*zS)/div*S) (may be out of range, reported as not ok, DS:50-51); R = Q[idx,act] + sigma*zR.
    """
    q = np.asarray(q, dtype=np.float64)
    S, A = q.shape
    g = np.arange(N, dtype=np.uint64) + np.uint64(offset)
    G, k = g >> np.uint64(2), (g & np.uint64(3)).astype(np.int64)
    words = []
    for c in range(3):
        ctr = np.uint64(3) * G + np.uint64(c)
        words.extend(philox4x32_10((ctr & np.uint64(0xFFFFFFFF)).astype(np.uint32), (ctr >> np.uint64(32)).astype(np.uint32),
                                   np.uint32(stream), np.uint32(0), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    w = np.stack(words, axis=1)                                   # (N, 12)
    rows = np.arange(N)
    x0, x1, x2 = w[rows, 3 * k], w[rows, 3 * k + 1], w[rows, 3 * k + 2]
    act = mulhi_u32(x0, A)
    z_r, z_s = box_muller(x1, x2)
    idx = np.floor((state_loc + state_scale * z_s) / state_div * S).astype(np.int64)
    ok = (idx >= 0) & (idx < S)
    r = np.where(ok, q[np.clip(idx, 0, S - 1), act] + sigma * z_r, 0.0)
    return idx, act, r, ok
