"""The two by-products of the online loop added with ABI 8 (csrc/diag.hip): the reference's third per-record trace
(true_step_TSRL_value, S1:96 / S2:94) from a kernel, and the top-2 gap census SURVEY.md section 7 asks for next to every parity run."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import c_oracle as co          # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def dc():
    import dcarl_amd
    dcarl_amd.require_gpu()
    return dcarl_amd


# ---- true_step_TSRL_value ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,S,A", [("sim1_trace.npz", 1, 30), ("sim2_trace.npz", 20, 11)])
@pytest.mark.parametrize("storage", ["f64", "f32"])
def test_true_step_values_equal_the_reference_trace(dc, golden, sim1_data, sim2_data, name, S, A, storage):
    """Q*[idx][TSRL_act] after every record, from dcarl_true_step_values_* over the kernel's own step_act, against the list the
    unmodified reference built (tests/golden/*_trace.npz true_step_value): EXACTLY for f64 storage (a gather of f64 values), the
    f32 rounding of the same values for f32 storage."""
    data, q = sim1_data if S == 1 else sim2_data
    g = golden(name)
    dt = torch.float64 if storage == "f64" else torch.float32
    table = dc.RecordTable.from_reference_table(data, S, A, storage=dt, limit=20000)
    tr = dc.ConfidenceEstimator().trace(table).check()
    ts = tr.true_step_values(q)
    assert ts.dtype == dt and ts.numel() == tr.step_val.numel()
    got = ts[table.state_major_index()].cpu().numpy()
    want = g["true_step_value"]
    if storage == "f64":
        assert np.array_equal(got, want)
    else:
        assert np.array_equal(got, want.astype(np.float32))
    # ... and in arrival order (what the script's progress print reads)
    arr = ts[table.rec_elem].double().cpu().numpy()
    sa = tr.steps_in_arrival_order()[1].cpu().numpy().astype(np.int64)
    st = table.rec_state.cpu().numpy()
    ref = np.asarray(q)[st, sa]
    assert np.array_equal(arr, ref if storage == "f64" else ref.astype(np.float32).astype(np.float64))


def test_drop_in_globals_carry_the_kernel_made_third_trace(dc, golden, sim2_data):
    from dcarl_amd import reference_api as api
    g2 = api.run_simulation(sim2_data[0], sim2_data[1], 20, 11, with_overall=True)
    ref = golden("sim2_trace.npz")
    flat = np.concatenate([np.asarray(x, dtype=np.float64) for x in g2["true_step_TSRL_value"]])
    assert np.array_equal(flat, ref["true_step_value"])


@pytest.mark.parametrize("S,A,maxlen,shared,sort", [(1, 3, 40, True, True), (70, 11, 300, False, True), (333, 16, 90, False, False),
                                                    (200, 24, 130, True, True), (129, 32, 77, False, True)])
def test_true_step_values_on_ragged_tables(dc, S, A, maxlen, shared, sort):
    """Ragged streams, sorted slots (the per-state Q row must follow the STATE, not the slot), a shared Q row, A up to 32, padding
    left at zero."""
    rng = np.random.RandomState(S + A)
    lens = rng.randint(0, maxlen + 1, S)
    lens[rng.randint(0, S)] = maxlen
    N = int(lens.sum())
    act = rng.randint(0, A, N).astype(np.uint8)
    q = rng.uniform(-50, 100, (1 if shared else S, A))
    st = np.repeat(np.arange(S), lens)
    R = (q[0 if shared else st, act] + 50.0 * rng.standard_normal(N)).astype(np.float32)
    table = dc.RecordTable.from_state_major(R, act, lens, A, sort_by_length=sort)
    tr = dc.ConfidenceEstimator().trace(table).check()
    ts = tr.true_step_values(q if not shared else q[0])
    idx = table.state_major_index()
    sa = tr.step_act[idx].cpu().numpy().astype(np.int64)
    want = (q[0][sa] if shared else q[st, sa]).astype(np.float32)
    assert np.array_equal(ts[idx].cpu().numpy(), want)
    mask = torch.ones(ts.numel(), dtype=torch.bool, device=ts.device)
    mask[idx] = False
    assert not ts[mask].any()                         # padding elements: zeros (what the buffer held)


def test_true_step_values_argument_errors(dc):
    R = np.zeros(10, np.float32)
    tbl = dc.RecordTable.from_state_major(R, np.zeros(10, np.uint8), np.array([10]), 3)
    est = dc.ConfidenceEstimator()
    with pytest.raises(ValueError):
        est.trace(tbl, want_steps=False).true_step_values(np.zeros((1, 3)))
    with pytest.raises(ValueError):
        est.trace(tbl).true_step_values(np.zeros((2, 3)))                    # neither one shared row nor one per state


# ---- top-2 gap census ------------------------------------------------------------------------------------------------------------------
def census_by_hand(V):
    """(evaluations, same-block count, histogram by bin) of a table of stripped values, NumPy: the definition of include/dcarl.h."""
    V = np.asarray(V, dtype=np.float64)
    bits = V.view(np.int64) & ~np.int64(31)
    s = np.sort(bits.view(np.float64), axis=1)
    hi, lo = s[:, -1], s[:, -2]
    same = int((hi.view(np.int64) == lo.view(np.int64)).sum())
    rel = (hi - lo) / np.maximum(np.abs(hi), 1e-300)
    e = ((rel.view(np.int64) >> 52) & 0x7ff) - 1023
    b = np.clip(e + 53, 0, 63)
    return V.shape[0], same, np.bincount(b, minlength=64), rel


def test_census_of_a_final_table_matches_numpy(dc):
    rng = np.random.RandomState(7)
    S, A = 5000, 11
    V = rng.uniform(-50, 100, (S, A))
    V[:100, 3] = V[:100, 7]                                   # exact ties between two candidates ...
    V[:100, [3, 7]] += 200.0                                  # ... that ARE the top two
    V[100:150, :] = -50.0                                     # all priors: true ties
    V[150:160, 2] = np.nextafter(V[150:160, 5] + 300, np.inf) # one ulp apart, on top
    V[150:160, 5] += 300
    est = dc.ConfidenceEstimator()
    rep = dc.census_report(est.top2_census(V=torch.from_numpy(V).cuda()))
    n, same, hist, rel = census_by_hand(V)
    assert rep["evaluations"] == n
    assert rep["same_32ulp_block"] == same >= 150
    assert rep["same_block_true_ties_at_prior"] == 50
    got = np.zeros(64, np.int64)
    for k, v in rep["log2_relative_gap_histogram"].items():
        got[int(k[2:]) + 53] = v
    assert np.array_equal(got, hist)
    stripped_same = (np.sort(V.view(np.int64) & ~np.int64(31), 1).view(np.float64))
    outside = rel[(stripped_same[:, -1].view(np.int64) != stripped_same[:, -2].view(np.int64))]
    assert rep["smallest_relative_gap_outside_window"] == outside.min()


@pytest.mark.parametrize("A,storage", [(11, "f32"), (16, "f32"), (3, "f64"), (24, "f32")])
def test_census_of_the_online_loop_counts_every_record(dc, A, storage):
    """One evaluation per record; the histogram equals the one NumPy makes from the C oracle's V after every record (the top two of
    the table as it stands), on a table small enough to replay on the host."""
    rng = np.random.RandomState(A)
    S, T = 70, 160
    lens = rng.randint(T // 2, T + 1, S)
    N = int(lens.sum())
    act = rng.randint(0, A, N).astype(np.uint8)
    q = rng.uniform(-50, 100, (S, A))
    st = np.repeat(np.arange(S), lens)
    dt = np.float32 if storage == "f32" else np.float64
    R = (q[st, act] + 50.0 * rng.standard_normal(N)).astype(dt)
    tbl = dc.RecordTable.from_state_major(R, act, lens, A, storage=torch.float32 if storage == "f32" else torch.float64)
    est = dc.ConfidenceEstimator()
    acc = est.top2_census(table=tbl)
    rep = dc.census_report(acc)
    assert rep["evaluations"] == N and rep["single_candidate_evaluations"] == 0
    assert sum(rep["log2_relative_gap_histogram"].values()) == N
    # replay on the host: the table after every record from prefix runs of the C oracle would be O(N^2); instead rebuild V step by step
    # from the oracle's own per-record outputs: V changes only at the record's (state, action), to the value the kernel's final-state
    # evaluation gives for that bucket prefix — taken from bounds_csr on the prefixes of ONE state (state 0 .. 2 only, to stay small)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    hist = np.zeros(64, np.int64)
    n_checked = 0
    for s in range(3):
        Rs, As = R[off[s]:off[s + 1]], act[off[s]:off[s + 1]]
        sub = dc.RecordTable.from_state_major(Rs, As, np.array([len(Rs)]), A, storage=tbl.R.dtype)
        one = dc.census_report(est.top2_census(table=sub))
        V = np.full(A, -50.0); V[0] = 100.0
        h = np.zeros(64, np.int64)
        for t in range(len(Rs)):
            pre = co.trace(Rs[:t + 1], As[:t + 1], np.array([0, t + 1], np.int64), 1, A, want_steps=False)
            _, _, hb, _ = census_by_hand(pre["V"])
            h += hb
        got = np.zeros(64, np.int64)
        for k, v in one["log2_relative_gap_histogram"].items():
            got[int(k[2:]) + 53] = v
        # the oracle's V and the kernel's agree to 1e-10, the bins are a factor 2 wide: a gap sitting on a bin edge may fall either side
        assert np.abs(got - h).sum() <= 2, (got, h)
        n_checked += len(Rs)
    assert n_checked > 0
    # accumulation: two launches into one accumulator add up
    acc2 = est.top2_census(table=tbl, into=est.top2_census(table=tbl))
    assert dc.census_report(acc2)["evaluations"] == 2 * N


def test_census_single_candidate_and_empty(dc):
    est = dc.ConfidenceEstimator()
    tbl = dc.RecordTable.from_state_major(np.ones(30, np.float32), np.zeros(30, np.uint8), np.array([30]), 1)
    rep = dc.census_report(est.top2_census(table=tbl))
    assert rep["evaluations"] == 30 and rep["single_candidate_evaluations"] == 30 and rep["same_32ulp_block"] == 0
    rep = dc.census_report(est.new_census())
    assert rep["evaluations"] == 0 and rep["smallest_relative_gap_outside_window"] is None


def test_sampler_straight_into_the_estimator_follows_the_visit_law_and_the_oracle(dc):
    """reference_api.Data_Generation_into_estimator: data_sampling.py's law drawn into the layout + the online loop, no table in between.
    The per-state record counts follow the visit law of DS:12-17 (chi-square against Phi-differences), the returns the N(Q*[a], 50) law,
    and the loop's outputs on the drawn table equal the C oracle's on the very same records."""
    from scipy import stats
    api = dc.reference_api
    api.seed(123)
    S, N, A = 20, 400_000, 11
    g, states, q = api.Data_Generation_into_estimator(S, N, A)
    lens = np.asarray(g["state_data_len"])
    edges = np.arange(S + 1) * 6.0 / S - 3.0
    p = np.diff(stats.norm.cdf(edges))
    assert abs(lens.sum() - N * p.sum()) < 6 * np.sqrt(N)                       # 0.27 % of the visits fall outside [0, S) and are dropped
    chi2 = (((lens - lens.sum() * p / p.sum()) ** 2) / (lens.sum() * p / p.sum())).sum()
    assert chi2 < stats.chi2.ppf(1 - 1e-6, S - 1)
    tbl, tr = g["table"], g["result"]
    idx = tbl.state_major_index()
    R, act = tbl.R[idx].cpu().numpy(), tbl.act[idx].cpu().numpy()
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    st = np.repeat(np.arange(S), lens)
    z = (R - q[st, act].astype(np.float32)) / 50.0
    assert stats.kstest(z[::7], "norm").pvalue > 1e-6 and np.bincount(act, minlength=A).min() > 0.95 * len(act) / A
    ref = co.trace(R, act, off, S, A)
    assert np.array_equal(np.concatenate([np.asarray(x, dtype=np.int64) for x in g["step_TSRL_act"]]), ref["step_act"])
    assert np.array_equal(g["activation_step"], ref["activation_step"]) and np.array_equal(tr.n.cpu().numpy(), ref["n"])
    flat_true = np.concatenate([np.asarray(x) for x in g["true_step_TSRL_value"]])
    assert np.array_equal(flat_true, q[st, ref["step_act"]].astype(np.float32).astype(np.float64))
