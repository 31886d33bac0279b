"""The reference's Python entry-point surface, backed by the HIP kernels.

Names, positional order, defaults and return types follow the reference verbatim
(S1 = Simulation_testing/Simulation_1/test_DCARL.py, S2 = .../Simulation_2/test_DCARL.py,
DS = Simulation_testing/Simulation_Data_Collection/Data_Sampling/data_sampling.py) so that the scripts of the
same names in this repo are drop-ins.  No CPU path: without a gfx950 GPU every function raises DcarlError."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .estimator import ConfidenceEstimator
from .params import Params
from .records import RecordTable
from . import sampler as _sampler


# ---- S1:10-28 ------------------------------------------------------------------------------------------
def _four_bounds(data_array, alpha, scale):
    x = np.ascontiguousarray(np.asarray(data_array, dtype=np.float64).ravel())
    if x.size == 0:
        raise ZeroDivisionError("float division by zero")      # what S1:12 raises for an empty bucket
    dev = _lib.require_gpu()
    lib = _lib.load()
    xv = torch.from_numpy(x).to(dev)
    off = torch.tensor([0, x.size], dtype=torch.int64, device=dev)
    out = torch.empty((1, 4), dtype=torch.float64, device=dev)
    p = Params(alpha=alpha, scale=scale).to_c()
    _lib.check(lib.dcarl_bucket_bounds_f64(_lib.ptr(xv), _lib.ptr(off), 1, C.byref(p), _lib.ptr(out),
                                           _lib.stream_ptr()), "dcarl_bucket_bounds_f64")
    return out[0].tolist()


def upper_bound(data_array, alpha=0.05, loc=-50, scale=150):
    """S1:10-12: min(100, mean + scale*sqrt(log(1/alpha)/2/n)).  ``loc`` is unused, as in the reference."""
    return _four_bounds(data_array, alpha, scale)[0]


def lower_bound(data_array, alpha=0.05, loc=-50, scale=150):
    """S1:14-16: mean - scale*sqrt(log(1/alpha)/2/n)."""
    return _four_bounds(data_array, alpha, scale)[1]


def CI_lower_bound(data_array, alpha=0.05, loc=-50, scale=150):
    """S1:18-24: dsum/n/(n+1) - 4*sigma/(n+1) + dsum/(n+1) - scale*sqrt(log(1/alpha)/2/(n+1))."""
    return _four_bounds(data_array, alpha, scale)[2]


def mean_value(data_array, alpha=0.05, loc=-50, scale=150):
    """S1:26-28: min(100, mean)."""
    return _four_bounds(data_array, alpha, scale)[3]


# ---- S1:31-111 / S2:30-135: the script bodies ---------------------------------------------------------------
def run_simulation(data, true_action_values, state_num, action_num, limit=20000, with_overall=False,
                   params: Params = Params(), storage=torch.float64, log_every=0, log=print):
    """Runs the online loop on the GPU and returns the reference's script-level globals (same names).

    ``data`` (N,4) float64 {state idx, state feature, action, cumulative reward}; only data[0:limit] is consumed
    (S1:73).  ``log_every`` reproduces the progress print of S1:101-102."""
    est = ConfidenceEstimator(params)
    table = RecordTable.from_reference_table(data, state_num, action_num, storage=storage, limit=limit)
    tr = est.trace(table).check()      # the results go to the host below: the one point where the launch's fault word is read
    sv_sm, sa_sm = tr.steps_by_state()
    sv_sm = sv_sm.to(torch.float64).cpu().numpy()
    sa_sm = sa_sm.cpu().numpy().astype(np.int64)
    lens = table.lengths_by_state.cpu().numpy().astype(np.int64)
    off = np.concatenate([[0], np.cumsum(lens)])
    q = np.asarray(true_action_values)
    step_TSRL_value = [sv_sm[off[s]:off[s + 1]].tolist() for s in range(state_num)]
    step_TSRL_act = [sa_sm[off[s]:off[s + 1]].tolist() for s in range(state_num)]
    # S1:96 / S2:94: true_action_values[idx][TSRL_act] per record — one gather kernel over the step_act trace (a state the
    # file has no row for would raise IndexError in the reference at its first record; here its list stays empty)
    if len(q) >= state_num:
        ts_sm = tr.true_step_values(q[:state_num])[table.state_major_index()].to(torch.float64).cpu().numpy()
        true_step_TSRL_value = [ts_sm[off[s]:off[s + 1]].tolist() for s in range(state_num)]
    else:
        true_step_TSRL_value = [q[s][sa_sm[off[s]:off[s + 1]]].tolist() if s < len(q) else [] for s in range(state_num)]
    g = dict(TSRL_value=tr.V.cpu().numpy().tolist(), step_TSRL_value=step_TSRL_value, step_TSRL_act=step_TSRL_act,
             true_step_TSRL_value=true_step_TSRL_value,
             activation_step=tr.activation_step.cpu().numpy().astype(np.int64),
             activation_value=np.array([-1] * state_num), state_data_len=lens.tolist(), k=table.n_records,
             data_state_act_len=tr.n.cpu().numpy())
    # S1:41,80: data_state_act[idx][act] = the bucket's rewards in arrival order (the final-state layout of the same table)
    vals, seg = table.to_buckets()
    vals = vals.to(torch.float64).cpu().numpy()
    seg = seg.cpu().numpy()
    g["data_state_act"] = [[vals[seg[s * action_num + a]:seg[s * action_num + a + 1]].tolist() for a in range(action_num)]
                           for s in range(state_num)]
    g["overall_value"] = []                                                          # S1:48: declared, never filled by S1
    if log_every:
        sv_k, sa_k = tr.steps_in_arrival_order()
        sv_k = sv_k.to(torch.float64).cpu().numpy()
        sa_k = sa_k.cpu().numpy()
        st_k = table.rec_state.cpu().numpy()
        is0 = np.flatnonzero(st_k == 0)
        for k in range(log_every, table.n_records + 1, log_every):      # S1:101-102
            j = np.searchsorted(is0, k - 1, side="right") - 1           # last arrival <= k-1 of state 0
            if j < 0:
                raise IndexError("list index out of range")             # step_TSRL_value[0][-1] on an empty list
            a0 = int(sa_k[is0[j]])
            log(k, int(sa_k[k - 1]), float(sv_k[is0[j]]), float(q[0][a0]))
    if with_overall:
        g["overall_value"] = est.overall_value(tr).cpu().numpy().tolist()          # S2:99-105
        arr = np.stack([np.arange(state_num), lens], axis=1)                         # S2:108-119
        g["sorted_state_data_len"] = arr[np.argsort(-lens)]
    return g


def plot_states(g, per_figure=5, plt=None):
    """The figures of S2:119-135: the states in descending order of visit count, ``per_figure`` stacked panels per
    figure (figure numbers 1, 2, ...), each panel the state's step-value trace — dark grey while the rule action is still
    the arg-max, black from the activation step on — on a common x-range (the longest trace).  Returns the figures."""
    if plt is None:
        import matplotlib.pyplot as plt
    order = np.asarray(g["sorted_state_data_len"])
    longest = int(order[0][1]) if len(order) else 0
    figures = []
    for rank, (sid, length) in enumerate(order.tolist()):
        panel = rank % per_figure
        if panel == 0:
            figures.append(plt.figure(rank // per_figure + 1))
        ax = plt.subplot(per_figure, 1, panel + 1)
        trace = g["step_TSRL_value"][sid]
        latch = int(g["activation_step"][sid])
        split = length if latch == -1 else latch
        ax.plot(trace[:split], color="darkgray")
        if latch != -1:
            ax.plot(range(latch, length), trace[latch:length], color="black")
        ax.set_xlim((0, longest))
    return figures


# ---- DS:5-67 ---------------------------------------------------------------------------------------------
# Two random sources.  LEGACY (the default, what makes these functions drop-ins): the draws come from the generators the
# reference itself draws from — NumPy's global RandomState (scipy's rvs with random_state=None) and Python's `random` — in the
# reference's draw order, and only the arithmetic runs on the GPU, through the bit-exact injected-noise kernels.  After
# `np.random.seed(s); random.seed(s)` the outputs equal the reference's bit for bit (tests/golden/sampler_seed*.npz).
# PHILOX (`legacy_streams=False`, or `use_legacy_streams(False)` for the whole module): the library's own counter RNG on
# the GPU, seeded with `seed()`; same distributions (KS / chi-square tests), not the same numbers — the fast path for large N.
_rng_state = {"seed": None, "calls": 0, "legacy": True}


def use_legacy_streams(flag: bool = True):
    """Module-wide default of ``legacy_streams`` (True: NumPy's global RandomState + Python's ``random``, like the reference)."""
    _rng_state["legacy"] = bool(flag)


def _legacy(flag):
    return _rng_state["legacy"] if flag is None else bool(flag)


def seed(value):
    """Seeds the Philox stream of the non-legacy path (the reference never seeds; DS has no equivalent)."""
    _rng_state["seed"] = int(value)
    _rng_state["calls"] = 0


def _next_stream():
    if _rng_state["seed"] is None:
        import os
        _rng_state["seed"] = int.from_bytes(os.urandom(8), "little")
    _rng_state["calls"] += 1
    return _rng_state["seed"], _rng_state["calls"]


def add_an_act_data(act, action_value, legacy_streams=None):
    """DS:5-9: one return sample ~ N(action_value[act], 50) as a Python float."""
    if _legacy(legacy_streams):
        z = np.random.standard_normal(1)                                   # norm.rvs(size=1): one legacy gauss draw
        rows, _ = _sampler.sample_from_noise([0.0], [[float(action_value[act])]], [0.0], [0], z)   # visit 0 -> state 0 (kept)
        return float(rows[0, 3].item())
    sd, call = _next_stream()
    q = torch.tensor([[float(action_value[act])]], dtype=torch.float32)    # a 1-candidate table: the draw is Q + 50 z
    tbl = _sampler.sample_state_records(q, 1, sd, stream_id=0x10000 + call)
    return float(tbl.R[tbl.state_major_index()][0].item())


def _standard_normals(n, sd, stream_id):
    """n float64 standard normals from the HIP sampler (Q = 0, sigma = 1 turns R into z)."""
    tbl = _sampler.sample_state_records(torch.zeros((1, 1), dtype=torch.float32), n, sd, sigma=1.0, stream_id=stream_id)
    return tbl.R[tbl.state_major_index()].to(torch.float64).cpu().numpy()


def random_state_norm(state_num, size, legacy_streams=None):
    """DS:12-17: floor(N(3,1)/6*state_num).astype(int); values may fall outside [0,state_num)."""
    if _legacy(legacy_streams):
        return _sampler.visit_floor(np.random.standard_normal(size), state_num).cpu().numpy().astype(int)
    sd, call = _next_stream()
    z = _standard_normals(size, sd, 0x20000 + call)
    return np.floor((3.0 + 1.0 * z) / 6 * state_num).astype(int)


def random_state_manual(state_num, size, legacy_streams=None):
    """DS:19-28: 10 % state 0, else uniform on 1..state_num-1 (defined, never called by Data_Generation)."""
    if _legacy(legacy_streams):
        import random
        u, r = [], []
        for _ in range(size):                                              # the reference's draw order (DS:22-24)
            x = random.random()
            u.append(x)
            if x > 0.1:
                r.append(random.randint(1, state_num - 1))
        return _sampler.state_manual_from_streams(u, r).cpu().tolist() if size else []
    sd, call = _next_stream()
    # the sampler's uniform action draw over k candidates is the uniform integer source: k = state_num-1 and k = 10
    pick = _sampler.sample_state_records(torch.zeros((1, max(1, state_num - 1))), size, sd, stream_id=0x30000 + call)
    coin = _sampler.sample_state_records(torch.zeros((1, 10)), size, sd, stream_id=0x40000 + call)
    a = pick.act[pick.state_major_index()].cpu().numpy().astype(int) + 1
    c = coin.act[coin.state_major_index()].cpu().numpy()
    return [int(x) if k != 0 else 0 for x, k in zip(a, c)]                 # "random.random() > 0.1" <=> coin != 0


def legacy_generation_streams(state_num=20, data_size=50000, action_num=11):
    """Every random draw of the reference's Data_Generation, from the generators it uses, in its order (DS:39, 43, 45, 54,
    55): uniform.rvs -> RandomState.random_sample, norm.rvs -> RandomState.standard_normal (scipy draws from NumPy's global
    RandomState), random.randint -> Python's ``random``.  The two generators are independent, so the per-visit pairs
    (randint, one normal) can be drawn as two runs.  Host code: this IS the reference's random source, not arithmetic."""
    import random
    u_states = np.random.random_sample(state_num)
    u_q = np.stack([np.random.random_sample(action_num) for _ in range(state_num)]) if state_num else np.zeros((0, action_num))
    z_visit = np.random.standard_normal(data_size)
    return u_states, u_q, z_visit, random, np.random.standard_normal


def Data_Generation(out_dir="Simulation_testing/Simulation_Data_Collection/", state_num=20, data_size=50000,
                    action_num=11, min_value=-50, max_value=100, legacy_streams=None):
    """DS:30-67: draws states, Q*, visits and returns and writes the three .npy files the reference writes (same relative
    paths, same dtypes/shapes).  With legacy streams (default) and ``np.random.seed(s); random.seed(s)`` beforehand the three
    files equal the reference's bit for bit."""
    if _legacy(legacy_streams):
        u_states, u_q, z_visit, pyrandom, normal = legacy_generation_streams(state_num, data_size, action_num)
        states = 0.0 + 1.0 * u_states                                                                     # DS:39 loc + scale*u
        action_values = min_value + (max_value - min_value) * u_q                                         # DS:42-43
        idx = _sampler.visit_floor(z_visit, state_num)                                                    # DS:45
        n_kept = int(((idx >= 0) & (idx < state_num)).sum().item())                                       # DS:50-51
        acts = [pyrandom.randint(0, action_num - 1) for _ in range(n_kept)]                               # DS:54
        z_reward = normal(n_kept)                                                                         # DS:55 (DS:9)
        rows, _ = _sampler.sample_from_noise(states, action_values, z_visit, acts if n_kept else [0], z_reward if n_kept else [0.0])
        data = rows.cpu().numpy()
    else:
        sd, call = _next_stream()
        _lib.require_gpu()
        gen = torch.Generator(device="cpu").manual_seed(sd & 0x7FFFFFFFFFFFFFFF)
        states = torch.rand(state_num, generator=gen, dtype=torch.float64).numpy()                       # DS:39
        action_values = (min_value + (max_value - min_value) *
                         torch.rand((state_num, action_num), generator=gen, dtype=torch.float64)).numpy()  # DS:42-43
        idx, act, R = _sampler.sample_pairs(torch.from_numpy(action_values), data_size, sd, stream_id=0x50000 + call)
        keep = idx >= 0                                                                                   # DS:50-51
        idx_k = idx[keep].cpu().numpy().astype(np.int64)
        data = np.stack([idx_k.astype(np.float64), states[idx_k], act[keep].cpu().numpy().astype(np.float64),
                         R[keep].to(torch.float64).cpu().numpy()], axis=1)                                # DS:55
    np.save(out_dir + "data.npy", data)                                                                   # DS:65-67
    np.save(out_dir + "action_value.npy", action_values)
    np.save(out_dir + "states.npy", states)
    return None


def Data_Generation_into_estimator(state_num=20, data_size=50000, action_num=11, min_value=-50, max_value=100, params: Params = Params(),
                                   want_steps: bool = True):
    """Both halves of the path in one call, for a pipeline that OWNS both: what ``Data_Generation()`` would write (DS:30-67) and what
    ``test_DCARL.py`` would then make of it (S1:73-99) — without the (N,4) float64 table in between, without its .npy round trip and
    without an ingest.  The visit law of DS:12-17,45-51 decides how many of the ``data_size`` visits every state keeps (independent
    Poisson counts: the multinomial up to its total), every state's records (DS:54-55: uniform action, N(Q*[a], 50) return) are drawn
    straight INTO the sliced layout (``dcarl_sample_state_records_ragged``, Philox counter (t, state)) and the online loop runs on it.

    Same LAW as the two scripts run one after the other (state / action / return distributions and the per-state arrival order
    semantics), not the same numbers (the library's Philox stream, ``seed()``), and the interleaving of the states' arrivals is never
    materialised — so there is no ``overall_value`` (S2:99-105 is the one output that reads it).  2.4 ms instead of 7.7 ms for 2^28
    records on 65 536 states (bench.py ``configs[2]->[1]``: the random arrival order costs the ingest's pack a 4x write amplification).

    Returns (globals dict as ``run_simulation``: TSRL_value, activation_step, state_data_len, step traces per state when ``want_steps``;
    states, action_values as Data_Generation draws them)."""
    from . import workloads as _wl
    sd, call = _next_stream()
    dev = _lib.require_gpu()
    gen = torch.Generator(device="cpu").manual_seed(sd & 0x7FFFFFFFFFFFFFFF)
    states = torch.rand(state_num, generator=gen, dtype=torch.float64).numpy()                                       # DS:39
    action_values = (min_value + (max_value - min_value) * torch.rand((state_num, action_num), generator=gen, dtype=torch.float64)).numpy()
    kept_share = 0.9973002039367398                                                                                  # P(|z| < 3): DS:50-51
    lengths = _wl.sim2_visit_lengths(state_num, mean=data_size * kept_share / state_num, seed=(sd ^ call) & 0x7FFFFFFF, device=dev)
    table = _sampler.sample_ragged_records(torch.from_numpy(action_values), lengths, seed=sd, stream_id=0x60000 + call)
    est = ConfidenceEstimator(params)
    tr = est.trace(table, want_steps=want_steps).check()
    lens = table.lengths_by_state.cpu().numpy().astype(np.int64)
    g = dict(TSRL_value=tr.V.cpu().numpy().tolist(), activation_step=tr.activation_step.cpu().numpy().astype(np.int64),
             state_data_len=lens.tolist(), k=table.n_records, data_state_act_len=tr.n.cpu().numpy(), table=table, result=tr)
    if want_steps:
        off = np.concatenate([[0], np.cumsum(lens)])
        idx = table.state_major_index()
        sv = tr.step_val[idx].to(torch.float64).cpu().numpy()
        sa = tr.step_act[idx].cpu().numpy().astype(np.int64)
        ts = tr.true_step_values(action_values)[idx].to(torch.float64).cpu().numpy()
        g["step_TSRL_value"] = [sv[off[s]:off[s + 1]].tolist() for s in range(state_num)]
        g["step_TSRL_act"] = [sa[off[s]:off[s + 1]].tolist() for s in range(state_num)]
        g["true_step_TSRL_value"] = [ts[off[s]:off[s + 1]].tolist() for s in range(state_num)]
    return g, states, action_values
