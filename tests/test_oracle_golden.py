"""Pin the CPU oracle against fixtures produced by the unmodified reference."""
import numpy as np
import pytest

from oracle import dcarl_oracle as orc


def test_bound_functions_bit_exact(golden):
    g = golden("bounds_random.npz")
    x, off = g["x"], g["off"]
    for i in range(len(off) - 1):
        b = x[off[i]:off[i + 1]]
        assert orc.upper_bound(b) == g["upper"][i]
        assert orc.lower_bound(b) == g["lower"][i]
        assert orc.CI_lower_bound(b) == g["ci_lower"][i]
        assert orc.mean_value(b) == g["mean_value"][i]
    b = x[off[5]:off[6]]
    for alpha, scale, ub, lb, ci in g["extra"]:
        assert orc.upper_bound(b, alpha, -50, scale) == ub
        assert orc.lower_bound(b, alpha, -50, scale) == lb
        assert orc.CI_lower_bound(b, alpha, -50, scale) == ci


def test_value_from_sums_matches_reference(golden):
    g = golden("bounds_random.npz")
    x, off = g["x"], g["off"]
    for i in range(len(off) - 1):
        b = x[off[i]:off[i + 1]]
        n, s, q = len(b), float(np.sum(b)), float(np.sum(b * b))
        tol = 1e-11 * max(1.0, abs(g["upper"][i]), float(np.max(np.abs(b))))
        assert abs(orc.value_from_sums(n, s, q, True) - g["upper"][i]) <= tol
        ref = min(g["lower"][i], g["ci_lower"][i])
        got = orc.value_from_sums(n, s, q, False)
        # q/n - mean^2 loses digits only for |mean| >> sigma (the 1e6 constant bucket)
        tol2 = 1e-11 * max(1.0, abs(ref)) + 4e-8 * float(np.max(np.abs(b))) * (float(np.std(b)) == 0)
        assert abs(got - ref) <= tol2, (i, got, ref)


def _check_trace(res, g, S):
    off = g["step_off"]
    for s in range(S):
        assert res["step_TSRL_value"][s] == list(g["step_value"][off[s]:off[s + 1]])
        assert [int(v) for v in res["step_TSRL_act"][s]] == list(g["step_act"][off[s]:off[s + 1]])
        assert res["true_step_TSRL_value"][s] == list(g["true_step_value"][off[s]:off[s + 1]])
    assert np.array_equal(np.array(res["TSRL_value"]), g["TSRL_value"])
    assert np.array_equal(res["activation_step"], g["activation_step"])
    assert np.array_equal(np.array(res["bucket_len"]), g["bucket_len"])


def test_sim1_faithful_bit_exact(golden, sim1_data):
    data, q = sim1_data
    g = golden("sim1_trace.npz")
    res = orc.run_online_faithful(data, 1, 30, true_action_values=q)
    _check_trace(res, g, 1)
    assert int(res["activation_step"][0]) == 4438
    assert res["step_TSRL_value"][0][-1] == 62.09592544287319


def test_sim2_faithful_bit_exact(golden, sim2_data):
    data, q = sim2_data
    g = golden("sim2_trace.npz")
    res = orc.run_online_faithful(data, 20, 11, true_action_values=q, with_overall=True)
    _check_trace(res, g, 20)
    assert np.array_equal(np.array(res["overall_value"]), g["overall_value"])
    assert res["overall_value"][-1] == 597.7193818873668
    lens = [len(t) for t in res["step_TSRL_value"]]
    assert np.array_equal(orc.trace_lengths_sorted(lens), g["sorted_state_data_len"])


@pytest.mark.parametrize("name,S,A,ov", [("sim1_trace.npz", 1, 30, False), ("sim2_trace.npz", 20, 11, True)])
def test_sums_formulation_matches_reference(golden, sim1_data, sim2_data, name, S, A, ov):
    data = (sim1_data if S == 1 else sim2_data)[0][:20000]
    g = golden(name)
    st, ac, r = data[:, 0].astype(int), data[:, 2].astype(int), data[:, 3]
    res = orc.run_online_sums(st, ac, r, S, A, with_overall=ov)
    # regroup arrival-order traces per state
    order = np.argsort(st, kind="stable")
    assert np.array_equal(res["step_act"][order], g["step_act"])           # arg-max bit-exact
    assert np.max(np.abs(res["step_val"][order] - g["step_value"])) <= 1e-11
    assert np.max(np.abs(res["V"] - g["TSRL_value"])) <= 1e-11
    assert np.array_equal(res["activation_step"], g["activation_step"])
    assert np.array_equal(res["n"], g["bucket_len"])
    if ov:
        assert np.max(np.abs(res["overall_value"] - g["overall_value"])) <= 1e-9


def test_bounds_batch_equals_final_table(golden, sim2_data):
    data = sim2_data[0][:20000]
    g = golden("sim2_trace.npz")
    st, ac, r = data[:, 0].astype(int), data[:, 2].astype(int), data[:, 3]
    key = st * 11 + ac
    order = np.argsort(key, kind="stable")
    seg = np.zeros(20 * 11 + 1, dtype=np.int64)
    seg[1:] = np.cumsum(np.bincount(key, minlength=220))
    V, n, vmax, amax = orc.bounds_batch(r[order], seg, 20, 11)
    assert np.array_equal(V, g["TSRL_value"])          # same samples, same order => bit-exact
    assert np.array_equal(n, g["bucket_len"])
    off = g["step_off"]
    last_act = np.array([g["step_act"][off[s + 1] - 1] for s in range(20)])
    assert np.array_equal(amax, last_act)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_sampler_restatement_bit_exact(golden, seed):
    g = golden(f"sampler_seed{seed}.npz")
    data, q, states = orc.data_generation_from_streams(g["u_states"], g["u_q"], g["z_visit"],
                                                       g["acts"], g["z_reward"])
    assert np.array_equal(data, g["data"])
    assert np.array_equal(q, g["action_value"])
    assert np.array_equal(states, g["states"])
    d2, q2, s2, _ = orc.data_generation_seeded(seed)
    assert np.array_equal(d2, g["data"]) and np.array_equal(q2, g["action_value"])
    assert np.array_equal(orc.random_state_norm_from_noise(20, g["random_state_norm_z"]),
                          g["random_state_norm_out"])


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_state_manual_restatement_and_legacy_draw_order(golden, seed):
    """a14: the restatement on the captured streams == the reference's output; and the host-side draw order of the drop-in's
    legacy mode (dcarl_amd.reference_api.legacy_generation_streams: NumPy's global RandomState + Python's random) reproduces
    the streams the reference consumed — including that n normals drawn in one call equal n single draws."""
    import random
    g = golden(f"sampler_seed{seed}.npz")
    assert orc.random_state_manual_from_streams(g["random_state_manual_u"], g["random_state_manual_r"]) == \
        g["random_state_manual_out"].tolist()
    assert abs((g["random_state_manual_out"] == 0).mean() - 0.1) < 0.03
    from dcarl_amd import reference_api as api
    np.random.seed(seed); random.seed(seed)
    u_states, u_q, z_visit, pyrandom, normal = api.legacy_generation_streams()
    assert np.array_equal(u_states, g["u_states"]) and np.array_equal(u_q, g["u_q"]) and np.array_equal(z_visit, g["z_visit"])
    idx = orc.random_state_norm_from_noise(20, z_visit)
    n_kept = int(((idx >= 0) & (idx < 20)).sum())
    assert [pyrandom.randint(0, 10) for _ in range(n_kept)] == g["acts"].tolist()
    assert np.array_equal(normal(n_kept), g["z_reward"])


def test_bundled_sampler_output_is_sim2_input(sim2_data):
    # Data_Sampling's bundled outputs are byte-identical to Simulation_2's inputs
    data, q = sim2_data
    assert data.shape == (49866, 4) and q.shape == (20, 11)
    assert data[:, 0].min() >= 0 and data[:, 0].max() <= 19
    assert set(np.unique(data[:, 2]).astype(int)) == set(range(11))


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, out in kat:
        got = orc.philox4x32_10(*ctr, *key)
        assert tuple(int(v) for v in got) == out


def test_own_sampler_distribution():
    rng = np.random.RandomState(3)
    q = rng.uniform(-50, 100, size=(20, 11))
    idx, act, r, ok = orc.sample_pairs(q, 200000, seed=7)
    assert 0.995 < ok.mean() < 0.9985                      # DS:50-51 drops ~0.27 %
    assert np.all(np.bincount(act, minlength=11) > 200000 / 11 * 0.95)
    resid = (r - q[np.clip(idx, 0, 19), act])[ok] / 50.0
    assert abs(resid.mean()) < 0.01 and abs(resid.std() - 1) < 0.01
    a2, r2, z2 = orc.sample_state_records(q, 5000, seed=7)
    assert a2.shape == (20, 5000) and abs(z2.mean()) < 0.01 and abs(z2.std() - 1) < 0.01


# ---- the two behaviours of the reference the HIP library REFUSES (tests/golden/refused_inputs.npz, made by the unmodified script) ------
def _check_refused(res, g, kind, S=20):
    off = g[f"{kind}_step_off"]
    for s in range(S):
        got = np.array(res["step_TSRL_value"][s], dtype=np.float64)
        assert np.array_equal(got, g[f"{kind}_step_value"][off[s]:off[s + 1]], equal_nan=True)
        assert [int(v) for v in res["step_TSRL_act"][s]] == list(g[f"{kind}_step_act"][off[s]:off[s + 1]])
    assert np.array_equal(np.array(res["TSRL_value"]), g[f"{kind}_TSRL_value"], equal_nan=True)
    assert np.array_equal(res["activation_step"], g[f"{kind}_activation_step"])
    assert np.array_equal(np.array(res["bucket_len"]), g[f"{kind}_bucket_len"])


def test_negative_state_ids_wrap_in_the_reference(golden):
    """S2:77-80: data_state_act[int(row[0])] with a negative id is Python indexing from the end — -1 files under state 19, -20 under
    state 0, silently.  The structure-faithful restatement (Python lists too) does what the reference did; the HIP library raises
    IndexError instead (tests/test_gpu_parity.py::test_refused_inputs_raise_where_the_reference_wraps_or_picks_nan)."""
    g = golden("refused_inputs.npz")
    data = g["negative_id_data"]
    assert (data[:, 0] < 0).sum() > 100
    res = orc.run_online_faithful(data, 20, 11)
    _check_refused(res, g, "negative_id")
    lens = g["negative_id_bucket_len"].sum(1)
    assert lens[19] == (data[:, 0] == -1).sum() and lens[0] == ((data[:, 0] == 0) | (data[:, 0] == -20)).sum()


def test_a_nan_reward_splits_max_and_argmax_in_the_reference(golden):
    """S2:91-92: once the bucket with the NaN passes the threshold its value is NaN; np.argmax returns the FIRST NaN for ever after
    (step_TSRL_act sticks to that candidate) while the builtin max() over the same row skips a NaN that is not its first element
    (step_TSRL_value keeps following the finite candidates): the reference's two traces disagree with each other from there on.
    The HIP library refuses non-finite rewards at the boundary (ValueError)."""
    g = golden("refused_inputs.npz")
    data = g["nan_reward_data"]
    s_nan, a_nan = (int(x) for x in g["nan_reward_bucket"])
    res = orc.run_online_faithful(data, 20, 11)
    _check_refused(res, g, "nan_reward")
    off = g["nan_reward_step_off"]
    acts = g["nan_reward_step_act"][off[s_nan]:off[s_nan + 1]]
    first = int(np.flatnonzero(acts == a_nan)[0])
    assert (acts[first:] == a_nan).all() and np.isnan(g["nan_reward_TSRL_value"][s_nan, a_nan])
    assert not np.isnan(g["nan_reward_step_value"]).any()
