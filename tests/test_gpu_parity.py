"""Parity tests proper: the HIP path (through the C-ABI) against the oracle and the reference goldens.

Tolerances (BASELINE.json north_star): values within 1e-5 relative for fp32 storage, arg-max bit-exact.
With f64 storage the kernels are held to 1e-10 against the reference itself."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import c_oracle as co          # noqa: E402  (checker only)
from oracle import dcarl_oracle as orc     # noqa: E402


def rel(a, b):
    return np.abs(a - b) / np.maximum(1.0, np.abs(b))


@pytest.fixture(scope="module")
def dc():
    import dcarl_amd
    dcarl_amd.require_gpu()
    return dcarl_amd


def group(data, S, limit=20000):
    d = data[:limit]
    st = d[:, 0].astype(np.int64)
    order = np.argsort(st, kind="stable")
    off = np.concatenate([[0], np.cumsum(np.bincount(st, minlength=S))]).astype(np.int64)
    return d[order, 3].copy(), d[order, 2].astype(np.uint8), off


# ---- a1-a4 ---------------------------------------------------------------------------------------------
def test_bound_functions_vs_reference_goldens(dc, golden):
    import ctypes as C
    from dcarl_amd import _lib
    g = golden("bounds_random.npz")
    lib = dc.load_library()
    dev = dc.require_gpu()
    off = torch.from_numpy(g["off"]).to(dev)
    B = len(g["off"]) - 1
    p = dc.Params().to_c()
    ref = np.stack([g["upper"], g["lower"], g["ci_lower"], g["mean_value"]], 1)
    for dt, fn, tol in ((torch.float64, lib.dcarl_bucket_bounds_f64, 1e-10), (torch.float32, lib.dcarl_bucket_bounds_f32, 1e-5)):
        x = torch.from_numpy(g["x"]).to(dev).to(dt)
        out = torch.zeros((B, 4), dtype=torch.float64, device=dev)
        _lib.check(fn(_lib.ptr(x), _lib.ptr(off), B, C.byref(p), _lib.ptr(out), _lib.stream_ptr()))
        got = out.cpu().numpy()
        # the 1e6-constant bucket loses digits in q/n - mean^2: judged on the value scale of that bucket
        scale = np.maximum(1.0, np.array([np.abs(g["x"][g["off"][i]:g["off"][i + 1]]).max() for i in range(B)]))
        err = np.abs(got - ref) / np.maximum(np.abs(ref), scale[:, None])
        assert err.max() <= tol, (dt, err.max(), np.unravel_index(err.argmax(), err.shape))


def test_drop_in_functions(dc, golden):
    g = golden("bounds_random.npz")
    api = dc.reference_api
    for i in (0, 17, 400, 900):
        x = g["x"][g["off"][i]:g["off"][i + 1]]
        assert abs(api.upper_bound(x) - g["upper"][i]) <= 1e-10 * max(1, abs(g["upper"][i]))
        assert abs(api.lower_bound(x) - g["lower"][i]) <= 1e-10 * max(1, abs(g["lower"][i]))
        assert abs(api.CI_lower_bound(x) - g["ci_lower"][i]) <= 1e-10 * max(1, abs(g["ci_lower"][i]))
        assert abs(api.mean_value(x) - g["mean_value"][i]) <= 1e-10 * max(1, abs(g["mean_value"][i]))
        assert isinstance(api.upper_bound(x), float)
    x = g["x"][g["off"][5]:g["off"][6]]
    for alpha, scale, ub, lb, ci in g["extra"]:
        assert abs(api.upper_bound(x, alpha, -50, scale) - ub) <= 1e-10 * max(1, abs(ub))
        assert abs(api.lower_bound(x, alpha, -50, scale) - lb) <= 1e-10 * max(1, abs(lb))
        assert abs(api.CI_lower_bound(x, alpha, -50, scale) - ci) <= 1e-10 * max(1, abs(ci))
    with pytest.raises(ZeroDivisionError):
        api.upper_bound(np.array([]))


# ---- a5-a10: the online loop on the bundled data ----------------------------------------------------------
@pytest.mark.parametrize("name,S,A", [("sim1_trace.npz", 1, 30), ("sim2_trace.npz", 20, 11)])
@pytest.mark.parametrize("storage", ["f64", "f32"])
def test_trace_bundled_data_vs_reference(dc, golden, sim1_data, sim2_data, name, S, A, storage):
    data = (sim1_data if S == 1 else sim2_data)[0]
    g = golden(name)
    dt = torch.float64 if storage == "f64" else torch.float32
    table = dc.RecordTable.from_reference_table(data, S, A, storage=dt, limit=20000)
    est = dc.ConfidenceEstimator()
    tr = est.trace(table)
    sv, sa = tr.steps_by_state()
    sv, sa = sv.double().cpu().numpy(), sa.cpu().numpy().astype(np.int64)
    assert np.array_equal(sa, g["step_act"])                               # arg-max bit-exact vs the reference
    tol = 1e-10 if storage == "f64" else 1e-5
    assert rel(sv, g["step_value"]).max() <= tol
    assert np.array_equal(tr.activation_step.cpu().numpy(), g["activation_step"])
    assert rel(tr.V.cpu().numpy(), g["TSRL_value"]).max() <= tol
    assert np.array_equal(tr.n.cpu().numpy(), g["bucket_len"])
    assert np.array_equal(tr.amax.cpu().numpy(), np.argmax(g["TSRL_value"], 1))
    if S == 20:
        ov = est.overall_value(tr).cpu().numpy()
        assert rel(ov, g["overall_value"]).max() <= (1e-9 if storage == "f64" else 1e-5)
        assert abs(ov[-1] - 597.7193818873668) <= (1e-8 if storage == "f64" else 6e-3)
    # same inputs through the C oracle (f32-rounded): exact arg-max, 1e-9 values
    R, act, off = group(data, S)
    ref = co.trace(R.astype(np.float32 if storage == "f32" else np.float64), act, off, S, A)
    assert np.array_equal(sa, ref["step_act"])
    assert rel(sv, ref["step_val"]).max() <= (1e-10 if storage == "f64" else 1e-6)


def test_drop_in_run_simulation_globals(dc, golden, sim1_data, sim2_data):
    api = dc.reference_api
    lines = []
    g1 = api.run_simulation(sim1_data[0], sim1_data[1], 1, 30, log_every=2000, log=lambda *a: lines.append(a))
    ref = golden("sim1_trace.npz")
    assert int(g1["activation_step"][0]) == 4438
    assert g1["step_TSRL_act"][0] == ref["step_act"].tolist()
    assert np.allclose(g1["true_step_TSRL_value"][0], ref["true_step_value"])
    assert len(lines) == 10 and lines[-1][0] == 20000 and lines[-1][1] == 1
    assert abs(lines[-1][2] - 62.09592544287319) < 1e-9 and lines[-1][3] == 67.6
    ref_lines = str(ref["stdout"]).strip().splitlines()
    for got, want in zip(lines, ref_lines[:10]):
        w = want.split()
        assert got[0] == int(w[0]) and got[1] == int(w[1]) and abs(got[2] - float(w[2])) < 1e-9 and got[3] == float(w[3])
    # S1:41,48,80: data_state_act = every bucket's rewards in arrival order (exactly the table's column 3), overall_value = []
    d1 = sim1_data[0][:20000]
    assert g1["overall_value"] == [] and len(g1["data_state_act"]) == 1 and len(g1["data_state_act"][0]) == 30
    for a in range(30):
        assert g1["data_state_act"][0][a] == d1[d1[:, 2] == a, 3].tolist(), a
    g2 = api.run_simulation(sim2_data[0], sim2_data[1], 20, 11, with_overall=True)
    d2 = sim2_data[0][:20000]
    for s_ in range(20):
        for a in range(11):
            assert g2["data_state_act"][s_][a] == d2[(d2[:, 0] == s_) & (d2[:, 2] == a), 3].tolist(), (s_, a)
    ref2 = golden("sim2_trace.npz")
    assert np.array_equal(g2["activation_step"], ref2["activation_step"])
    assert abs(g2["overall_value"][-1] - 597.7193818873668) < 1e-8
    assert np.array_equal(g2["sorted_state_data_len"], ref2["sorted_state_data_len"])


# ---- random ragged inputs vs the C oracle -----------------------------------------------------------------
@pytest.mark.parametrize("S,A,maxlen,seed", [(1, 11, 300, 0), (70, 1, 90, 8), (70, 2, 120, 9), (129, 24, 260, 10), (63, 3, 50, 1), (64, 8, 200, 2), (65, 9, 257, 3),
                                             (1000, 11, 400, 4), (777, 16, 123, 5), (300, 17, 90, 6), (130, 32, 500, 7),
                                             (200, 13, 333, 11), (321, 15, 210, 12), (90, 14, 77, 13), (150, 12, 260, 14)])
@pytest.mark.parametrize("mapping", ["default", "default-f64", "unsorted", "slices2", "slices3-f64", "slices4", "duo", "trio", "quad", "single", "single-f64"])
def test_trace_random_ragged_vs_oracle(dc, knob, S, A, maxlen, seed, mapping):
    """Every online kernel against the C oracle.  default: four (f32) / three (f64) waves per slice on round-robin quads sharing the
    count-root table for A <= 16, the one-wave compute kernel above; `duo` / `trio`: the two- / three-wave instances
    (fp32, A = 11 / 16; the default elsewhere); `quad`: four waves (the f32 default); `single`: the compute kernel everywhere; `slicesN`: N slices per workgroup
    pinned (the launcher's own choice for tables this small is 1; 4 is what 65 536 states and more run with)."""
    mapping, _, f64 = mapping.partition("-")
    unsorted = mapping == "unsorted"            # slots = states; everywhere else tables of more than 64 states carry sorted slots
    if mapping.startswith("slices"):
        knob("DCARL_TRACE_SLICES", mapping[6:])
    elif mapping not in ("default", "unsorted"):
        knob("DCARL_TRACE_KERNEL", mapping)
    rng = np.random.RandomState(seed)
    lens = rng.randint(0, maxlen + 1, S)
    lens[rng.randint(0, S)] = 0
    lens[rng.randint(0, S)] = maxlen
    N = int(lens.sum())
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    act = rng.randint(0, A, N).astype(np.uint8)
    q = rng.uniform(-50, 100, (S, A))
    sig = np.where(rng.rand(S) < 0.3, 0.2, 50.0)
    st = np.repeat(np.arange(S), lens)
    R = (q[st, act] + sig[st] * rng.standard_normal(N)).astype(np.float64 if f64 else np.float32)
    table = dc.RecordTable.from_state_major(R, act, lens, A, storage=torch.float64 if f64 else torch.float32,
                                            sort_by_length=not unsorted)
    assert (table.slot_state is not None) == (S > 64 and not unsorted)
    tr = dc.ConfidenceEstimator().trace(table)
    want = "trace_kernel<" if (mapping == "single" or A > 16) else "trace_nwave_kernel<"
    assert dc._lib.last_kernel().startswith(want + ("double" if f64 else "float")), dc._lib.last_kernel()
    sv, sa = tr.steps_by_state()
    ref = co.trace(R, act, off, S, A)
    assert np.array_equal(sa.cpu().numpy(), ref["step_act"])
    assert rel(sv.double().cpu().numpy(), ref["step_val"]).max() <= (1e-10 if f64 else 1e-6)   # f32 rounding of the trace output
    if f64:                                   # the tie-break code bits never reach the caller: priors come back exact
        got = sv.cpu().numpy()
        assert np.array_equal(got[ref["step_val"] == 100.0], np.full(int((ref["step_val"] == 100.0).sum()), 100.0))
        assert not (got.view(np.int64) & 31).any()
    assert np.array_equal(tr.activation_step.cpu().numpy(), ref["activation_step"])
    assert rel(tr.V.cpu().numpy(), ref["V"]).max() <= 1e-10
    assert np.array_equal(tr.n.cpu().numpy(), ref["n"])
    assert np.array_equal(tr.amax.cpu().numpy(), ref["amax"])
    assert rel(tr.vmax.double().cpu().numpy(), ref["vmax"].astype(np.float64)).max() <= 1e-6


@pytest.mark.parametrize("A,T,ragged", [(2, 12000, False), (3, 14000, True), (14, 9000, False)])
def test_trace_long_buckets_leave_the_count_table(dc, A, T, ragged):
    """Bucket counts beyond the 4096-entry count-root table: a wavefront must switch to the compute path for the
    rest of its stream (and the two paths must agree bit for bit, which the exact arg-max comparison checks)."""
    rng = np.random.RandomState(A)
    S = 130
    lens = rng.randint(T // 2, T + 1, S) if ragged else np.full(S, T)
    N = int(lens.sum())
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    pa = np.full(A, 0.1 / max(A - 1, 1)); pa[0] = 0.9 if A > 1 else 1.0       # one dominant bucket: n >> 4096
    act = rng.choice(A, size=N, p=pa / pa.sum()).astype(np.uint8)
    q = rng.uniform(-50, 100, (S, A))
    st = np.repeat(np.arange(S), lens)
    R = (q[st, act] + 50.0 * rng.standard_normal(N)).astype(np.float32)
    tr = dc.ConfidenceEstimator().trace(dc.RecordTable.from_state_major(R, act, lens, A))
    sv, sa = tr.steps_by_state()
    ref = co.trace(R, act, off, S, A)
    assert int(ref["n"].max()) > 4096 + 1000
    assert np.array_equal(sa.cpu().numpy(), ref["step_act"])
    assert rel(sv.double().cpu().numpy(), ref["step_val"]).max() <= 1e-6
    assert np.array_equal(tr.activation_step.cpu().numpy(), ref["activation_step"])
    assert rel(tr.V.cpu().numpy(), ref["V"]).max() <= 1e-10
    assert np.array_equal(tr.n.cpu().numpy(), ref["n"])


@pytest.mark.parametrize("A,storage", [(11, "f32"), (12, "f32"), (5, "f64"), (14, "f32"), (20, "f32")])
def test_trace_without_step_outputs_matches_with(dc, A, storage):
    """The instances without step-trace stores (want_steps=False) of every online kernel give the same final table,
    counts, arg-max and activation step as the instances with them."""
    rng = np.random.RandomState(100 + A)
    S, T = 200, 700
    lens = rng.randint(T - 90, T + 1, S)
    N = int(lens.sum())
    act = rng.randint(0, A, N).astype(np.uint8)
    st = np.repeat(np.arange(S), lens)
    q = rng.uniform(-50, 100, (S, A))
    R = (q[st, act] + 50.0 * rng.standard_normal(N)).astype(np.float32 if storage == "f32" else np.float64)
    tbl = dc.RecordTable.from_state_major(R, act, lens, A, storage=torch.float32 if storage == "f32" else torch.float64)
    est = dc.ConfidenceEstimator()
    a, b = est.trace(tbl), est.trace(tbl, want_steps=False)
    assert b.step_val is None and a.step_val is not None
    assert torch.equal(a.V, b.V) and torch.equal(a.n, b.n) and torch.equal(a.amax, b.amax)
    assert torch.equal(a.activation_step, b.activation_step) and torch.equal(a.vmax, b.vmax)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ref = co.trace(R, act, off, S, A)
    assert rel(b.V.cpu().numpy(), ref["V"]).max() <= 1e-10
    assert np.array_equal(b.amax.cpu().numpy(), ref["amax"]) and np.array_equal(b.n.cpu().numpy(), ref["n"])
    assert np.array_equal(b.activation_step.cpu().numpy(), ref["activation_step"])


def test_trace_ragged_reference_table_sorted_slots(dc):
    """An arrival-ordered (N,4) table with 300 ragged states: slots are the states sorted by stream length; every
    per-state output, the arrival-order traces and overall_value must come back in STATE / ARRIVAL order."""
    rng = np.random.RandomState(21)
    S, A, N = 300, 7, 60000
    pst = rng.dirichlet(np.full(S, 0.3))
    st = rng.choice(S, size=N, p=pst)
    st[st == 5] = 6                                               # state 5 never visited
    act = rng.randint(0, A, N)
    q = rng.uniform(-50, 100, (S, A))
    R = (q[st, act] + 50 * rng.standard_normal(N)).astype(np.float32).astype(np.float64)
    data = np.stack([st.astype(np.float64), rng.rand(N), act.astype(np.float64), R], 1)
    table = dc.RecordTable.from_reference_table(data, S, A, storage=torch.float32)
    assert table.state_slot is not None
    lens = np.bincount(st, minlength=S)
    assert np.array_equal(table.lengths_by_state.cpu().numpy(), lens)
    assert np.all(np.diff(table.lengths.cpu().numpy()) <= 0)      # slot order = descending length
    est = dc.ConfidenceEstimator()
    tr = est.trace(table)
    order = np.argsort(st, kind="stable")
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ref = co.trace(R[order].astype(np.float32), act[order].astype(np.uint8), off, S, A)
    sv, sa = tr.steps_by_state()
    assert np.array_equal(sa.cpu().numpy(), ref["step_act"])
    assert rel(sv.double().cpu().numpy(), ref["step_val"]).max() <= 1e-6
    assert np.array_equal(tr.activation_step.cpu().numpy(), ref["activation_step"])
    assert np.array_equal(tr.n.cpu().numpy(), ref["n"]) and np.array_equal(tr.amax.cpu().numpy(), ref["amax"])
    assert rel(tr.V.cpu().numpy(), ref["V"]).max() <= 1e-10
    svk, sak = tr.steps_in_arrival_order()
    pos = np.empty(N, np.int64)
    pos[order] = np.arange(N)
    assert np.array_equal(sak.cpu().numpy(), ref["step_act"][pos])
    ov = est.overall_value(tr).cpu().numpy()
    ov_ref = co.overall(ref["step_val"], ref["activation_step"], off, st.astype(np.int32), pos)
    assert rel(ov, ov_ref).max() <= 1e-5
    tr2 = est.trace(table, out=tr)                                # buffer reuse on a sorted table
    assert torch.equal(tr2.activation_step, tr.activation_step) and torch.equal(tr2.amax, tr.amax)


def test_trace_params_and_ties(dc):
    # non-default parameters, rule action != 0, exact ties (constant rewards) -> first arg-max wins
    rng = np.random.RandomState(11)
    S, A, T = 70, 6, 80
    lens = np.full(S, T)
    act = rng.randint(0, A, S * T).astype(np.uint8)
    R = np.full(S * T, 5.0, np.float32)
    off = np.arange(S + 1, dtype=np.int64) * T
    p = dc.Params(rule_act=2, n_thres=3, alpha=0.1, scale=40.0, cap=30.0, init_rule=30.0, init_other=-5.0)
    table = dc.RecordTable.from_state_major(R, act, lens, A)
    tr = dc.ConfidenceEstimator(p).trace(table)
    sv, sa = tr.steps_by_state()
    ref = co.trace(R, act, off, S, A, p=co.params(2, 3, 0.1, 40.0, 30.0, 30.0, -5.0))
    assert np.array_equal(sa.cpu().numpy(), ref["step_act"])
    assert rel(sv.double().cpu().numpy(), ref["step_val"]).max() <= 1e-6
    assert np.array_equal(tr.activation_step.cpu().numpy(), ref["activation_step"])


def test_trace_empty_and_steps_optional(dc):
    table = dc.RecordTable.from_state_major(np.zeros(0, np.float32), np.zeros(0, np.uint8), np.zeros(5, np.int64), 11)
    tr = dc.ConfidenceEstimator().trace(table, want_steps=False)
    assert tr.step_val is None
    assert tr.activation_step.cpu().tolist() == [-1] * 5
    assert tr.amax.cpu().tolist() == [0] * 5 and tr.vmax.cpu().tolist() == [100.0] * 5
    assert np.array_equal(tr.V.cpu().numpy(), np.array([[100.0] + [-50.0] * 10] * 5))


# ---- final-state kernel --------------------------------------------------------------------------------------
@pytest.mark.parametrize("S,A,nmean,seed", [(50, 11, 3, 0), (200, 11, 91, 1), (33, 16, 64, 2), (20, 11, 1818, 3),
                                            (500, 5, 20, 4), (7, 32, 300, 5)])
@pytest.mark.parametrize("storage", ["f32", "f64"])
def test_bounds_csr_vs_oracle(dc, S, A, nmean, seed, storage):
    rng = np.random.RandomState(seed)
    n = rng.poisson(nmean, S * A)
    n[rng.randint(0, S * A, 5)] = 0
    seg = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
    q = rng.uniform(-50, 100, S * A)
    vals = (np.repeat(q, n) + 50 * rng.standard_normal(int(seg[-1])))
    npdt = np.float32 if storage == "f32" else np.float64
    vals = vals.astype(npdt)
    dev = dc.require_gpu()
    pad = np.zeros(max(4, len(vals)), npdt)
    pad[:len(vals)] = vals
    res = dc.ConfidenceEstimator().bounds(torch.from_numpy(pad).to(dev), S, A, seg_off=torch.from_numpy(seg))
    ref = co.bounds_csr(vals if len(vals) else pad, seg, S, A)
    assert rel(res.V.cpu().numpy(), ref["V"]).max() <= 1e-10
    assert np.array_equal(res.n.cpu().numpy(), ref["n"])
    assert np.array_equal(res.amax.cpu().numpy(), ref["amax"])
    assert rel(res.vmax.double().cpu().numpy(), ref["vmax"].astype(np.float64)).max() <= 1e-6


def test_bounds_dense_and_consistency_with_trace(dc):
    # dense buckets (cfg5 shape: n=64, A=16) == CSR with uniform offsets; trace's final table == bounds on the same data
    rng = np.random.RandomState(3)
    S, A, n = 257, 16, 64
    vals = (rng.uniform(-50, 100, (S, A, 1)) + 50 * rng.standard_normal((S, A, n))).astype(np.float32)
    dev = dc.require_gpu()
    est = dc.ConfidenceEstimator()
    dense = est.bounds(torch.from_numpy(vals.ravel()).to(dev), S, A, n_dense=n)
    seg = np.arange(S * A + 1, dtype=np.int64) * n
    ref = co.bounds_csr(vals.ravel(), seg, S, A)
    assert rel(dense.V.cpu().numpy(), ref["V"]).max() <= 1e-10
    assert np.array_equal(dense.amax.cpu().numpy(), ref["amax"])
    # feed the same samples through the online kernel in a shuffled arrival order
    act = np.tile(np.repeat(np.arange(A), n), S).astype(np.uint8)
    perm = np.concatenate([rng.permutation(A * n) + s * A * n for s in range(S)])
    table = dc.RecordTable.from_state_major(vals.ravel()[perm], act[perm], np.full(S, A * n), A)
    tr = est.trace(table, want_steps=False)
    assert rel(tr.V.cpu().numpy(), dense.V.cpu().numpy()).max() <= 1e-10
    assert np.array_equal(tr.amax.cpu().numpy(), dense.amax.cpu().numpy())


def test_bounds_from_reference_table(dc, golden, sim2_data):
    g = golden("sim2_trace.npz")
    res = dc.ConfidenceEstimator().bounds_from_reference_table(sim2_data[0], 20, 11, storage=torch.float64, limit=20000)
    assert rel(res.V.cpu().numpy(), g["TSRL_value"]).max() <= 1e-10
    assert np.array_equal(res.n.cpu().numpy(), g["bucket_len"])
    assert np.array_equal(res.amax.cpu().numpy(), np.argmax(g["TSRL_value"], 1))


# ---- sampler ---------------------------------------------------------------------------------------------------
def test_sampler_state_records_vs_oracle(dc):
    rng = np.random.RandomState(5)
    S, A, T = 130, 11, 203
    q = rng.uniform(-50, 100, (S, A)).astype(np.float32)
    tbl = dc.sampler.sample_state_records(torch.from_numpy(q), T, seed=0x123456789ABCDEF, stream_id=3)
    idx = tbl.state_major_index()
    act = tbl.act[idx].cpu().numpy().reshape(S, T)
    R = tbl.R[idx].cpu().numpy().reshape(S, T)
    a_ref, r_ref = co.sample_state_records(q.astype(np.float64), T, seed=0x123456789ABCDEF, stream=3)
    assert np.array_equal(act, a_ref)                                   # Philox words + action map bit-exact
    # f32 Box-Muller (v_log_f32 / v_cos_f32 on a 24-bit u) vs float64 libm: |dz| <= 1e-4 worst case (tiny radii),
    # typically 1e-6; R = Q + 50 z
    assert np.abs(R - r_ref).max() <= 5e-3 and np.quantile(np.abs(R - r_ref), 0.999) <= 5e-4
    z = (R - q[np.arange(S)[:, None], act]) / 50.0
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02
    # shared Q row
    tb2 = dc.sampler.sample_state_records(torch.from_numpy(q[:1]), T, seed=9, S=70)
    assert tb2.S == 70 and tb2.n_records == 70 * T


def test_sampler_pairs_vs_oracle(dc):
    rng = np.random.RandomState(6)
    q = rng.uniform(-50, 100, (20, 11)).astype(np.float32)
    N = 200000
    idx, act, R = dc.sampler.sample_pairs(torch.from_numpy(q), N, seed=77, offset=(1 << 32) - 1000)
    i_ref, a_ref, r_ref = co.sample_pairs(q.astype(np.float64), N, seed=77, offset=(1 << 32) - 1000)
    idx, act, R = idx.cpu().numpy(), act.cpu().numpy(), R.cpu().numpy()
    assert np.array_equal(act, a_ref)
    same = idx == i_ref
    assert same.mean() > 0.9999                                         # the f32 NORMAL differs from libm's by ~1e-6: a visit that
    d = np.abs(R[same] - r_ref[same])                                   # close to a bin edge may land next door (see the test below)
    assert d.max() <= 5e-3 and np.quantile(d, 0.999) <= 5e-4
    for off, n in ((0, 7), (1, 9), (6, 1), (3, 258)):                   # unaligned offsets, partial groups
        i2, a2, r2 = dc.sampler.sample_pairs(torch.from_numpy(q), n, seed=77, offset=off)
        ir, ar, rr = co.sample_pairs(q.astype(np.float64), n, seed=77, offset=off)
        assert np.array_equal(a2.cpu().numpy(), ar) and np.abs(r2.cpu().numpy() - rr).max() <= 5e-3
    keep = idx >= 0
    assert 0.995 < keep.mean() < 0.9985                                 # DS:50-51 drops ~0.27 %
    hist = np.bincount(idx[keep], minlength=20) / keep.sum()
    exp = np.bincount(i_ref[i_ref >= 0], minlength=20) / (i_ref >= 0).sum()
    assert np.abs(hist - exp).max() < 1e-4


@pytest.mark.parametrize("N,S,offset", [(1_000_000, 20, 0), (1_000_000, 20, 3), (300_000, 1 << 20, 7), (200_000, 1, 0), (100_003, 333, 1)])
def test_sampler_pairs_index_is_exact_at_configs2_size(dc, N, S, offset):
    """configs[2] at its stated size (1e6 pairs, 20 states): the visit index is INDEX work, so it is held to bit-exactness —
    idx == floor((3 + 1*z)/6*S) in float64 (NumPy, the expression of DS:14-15) on EVERY draw, for the f32 normal z the kernel
    drew and hands out; the normal itself is floating-point work and is held to the Box-Muller tolerance against the oracle's
    float64 libm normal; actions (integer work) are bit-exact, returns within the same tolerance wherever the two visits agree."""
    rng = np.random.RandomState(S % 1000)
    q = rng.uniform(-50, 100, (S, 11)).astype(np.float32)
    idx, act, R, z = dc.sampler.sample_pairs(torch.from_numpy(q), N, seed=5, offset=offset, want_z=True)
    idx, act, R, z = idx.cpu().numpy(), act.cpu().numpy(), R.cpu().numpy(), z.cpu().numpy()
    v = np.floor((3.0 + 1.0 * z.astype(np.float64)) / 6 * S)                     # DS:14-15 verbatim, float64
    want = np.where((v < 0) | (v >= S), -1, v).astype(np.int32)                   # DS:50-51
    assert np.array_equal(idx, want)                                              # every draw
    i_ref, a_ref, r_ref, z_ref = co.sample_pairs(q.astype(np.float64), N, seed=5, offset=offset, want_z=True)
    assert np.array_equal(act, a_ref)
    assert np.abs(z - z_ref).max() <= 1e-4 and np.quantile(np.abs(z - z_ref), 0.999) <= 1e-5
    same = idx == i_ref
    assert same.mean() > (0.9999 if S <= 1000 else 0.5)                           # (2^20 bins are 6e-6 wide: next door is the norm)
    near = np.abs(idx.astype(np.int64) - i_ref) <= np.maximum(1, int(S * 2e-5))
    assert near[(idx >= 0) & (i_ref >= 0)].all()
    both = same & (idx >= 0)
    assert np.abs(R[both] - r_ref[both]).max() <= 5e-3
    # without the extra output: the same draws
    i2, a2, r2 = dc.sampler.sample_pairs(torch.from_numpy(q), N, seed=5, offset=offset)
    assert np.array_equal(i2.cpu().numpy(), idx) and np.array_equal(a2.cpu().numpy(), act) and np.array_equal(r2.cpu().numpy(), R)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_sampler_from_noise_bit_exact_vs_reference(dc, golden, seed):
    g = golden(f"sampler_seed{seed}.npz")
    states = 0.0 + 1.0 * g["u_states"]
    q = -50.0 + 150.0 * g["u_q"]
    assert np.array_equal(q, g["action_value"]) and np.array_equal(states, g["states"])
    out, idx = dc.sampler.sample_from_noise(states, q, g["z_visit"], g["acts"], g["z_reward"])
    assert np.array_equal(out.cpu().numpy(), g["data"])                 # bit-exact with the reference's data.npy
    ref_idx = np.floor((3.0 + g["z_visit"]) / 6 * 20).astype(int)
    ref_idx[(ref_idx < 0) | (ref_idx >= 20)] = -1
    assert np.array_equal(idx.cpu().numpy(), ref_idx)


def test_data_generation_drop_in(dc, tmp_path, monkeypatch):
    api = dc.reference_api
    monkeypatch.chdir(tmp_path)
    (tmp_path / "Simulation_testing" / "Simulation_Data_Collection").mkdir(parents=True)
    api.seed(123)
    assert api.Data_Generation(legacy_streams=False) is None              # the library's own Philox stream
    base = "Simulation_testing/Simulation_Data_Collection/"
    data, q, states = np.load(base + "data.npy"), np.load(base + "action_value.npy"), np.load(base + "states.npy")
    assert data.dtype == np.float64 and data.shape[1] == 4 and 49700 < data.shape[0] < 49950
    assert q.shape == (20, 11) and states.shape == (20,) and q.min() >= -50 and q.max() <= 100
    s, a = data[:, 0].astype(int), data[:, 2].astype(int)
    assert np.array_equal(data[:, 1], states[s])
    z = (data[:, 3] - q[s, a]) / 50
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02
    from scipy import stats
    assert stats.kstest(z, "norm").pvalue > 1e-3
    # and it is a valid Sim2 input
    g = api.run_simulation(data, q, 20, 11, with_overall=True)
    assert len(g["overall_value"]) == 20000
    for legacy in (False, True):
        r = api.random_state_norm(20, 1000, legacy_streams=legacy)
        assert r.dtype.kind == "i" and 8 < r.mean() < 11
        assert isinstance(api.add_an_act_data(3, q[0], legacy_streams=legacy), float)
        m = api.random_state_manual(20, 500, legacy_streams=legacy)
        assert isinstance(m, list) and len(m) == 500 and 0 <= min(m) and max(m) <= 19


def test_drop_in_scripts_run_from_repo_root(dc, golden, tmp_path):
    """The three scripts are launched exactly like the reference's (python <script> from the repo root)."""
    import os, shutil, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MPLBACKEND="Agg")
    out = subprocess.run([sys.executable, "Simulation_testing/Simulation_1/test_DCARL.py"], cwd=repo, env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    got = [l.split() for l in out.stdout.strip().splitlines()]
    want = [l.split() for l in str(golden("sim1_trace.npz")["stdout"]).strip().splitlines()]
    assert len(got) == len(want) == 11                        # ten progress lines (S1:101-102) + activation step
    for g, w in zip(got[:10], want[:10]):
        assert g[0] == w[0] and g[1] == w[1] and abs(float(g[2]) - float(w[2])) < 1e-9 and float(g[3]) == float(w[3])
    assert got[10] == want[10] == ["4438"]                    # S1:107
    out = subprocess.run([sys.executable, "Simulation_testing/Simulation_2/test_DCARL.py"], cwd=repo, env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip() == "", out.stderr[-2000:]     # Sim2 prints nothing
    # the sampler script writes into <cwd>/Simulation_testing/Simulation_Data_Collection/ (DS:65-67): run it in a copy
    work = tmp_path / "w"
    (work / "Simulation_testing" / "Simulation_Data_Collection" / "Data_Sampling").mkdir(parents=True)
    shutil.copy(os.path.join(repo, "Simulation_testing/Simulation_Data_Collection/Data_Sampling/data_sampling.py"),
                work / "Simulation_testing/Simulation_Data_Collection/Data_Sampling/data_sampling.py")
    env2 = dict(env, PYTHONPATH=repo)
    out = subprocess.run([sys.executable, "Simulation_testing/Simulation_Data_Collection/Data_Sampling/data_sampling.py"],
                         cwd=work, env=env2, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = np.load(work / "Simulation_testing/Simulation_Data_Collection/data.npy")
    assert d.shape[1] == 4 and 49700 < d.shape[0] < 49950
    assert np.load(work / "Simulation_testing/Simulation_Data_Collection/action_value.npy").shape == (20, 11)
    assert np.load(work / "Simulation_testing/Simulation_Data_Collection/states.npy").shape == (20,)


def test_integration_md_stub_runs_verbatim(dc, golden):
    """The ctypes stub INTEGRATION.md shows a reference maintainer is executed as written (docs cannot rot)."""
    import os
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(repo, "INTEGRATION.md")).read()
    start = text.index("```python") + len("```python")
    code = text[start:text.index("```", start)]
    cwd = os.getcwd()
    os.chdir(repo)
    try:
        ns = {}
        exec(compile(code, "INTEGRATION.md", "exec"), ns)
    finally:
        os.chdir(cwd)
    torch.cuda.synchronize()
    g = golden("sim1_trace.npz")
    assert int(ns["activation_step"][0]) == 4438
    assert np.array_equal(ns["step_TSRL_act"].cpu().numpy(), g["step_act"])
    assert np.abs(ns["step_TSRL_value"].cpu().numpy() - g["step_value"]).max() < 1e-9


def test_out_of_range_ids_raise_like_the_reference(dc):
    bad_state = np.array([[0, 0.5, 1, 3.0], [7, 0.5, 1, 3.0]])
    with pytest.raises(IndexError):
        dc.RecordTable.from_reference_table(bad_state, 5, 11)
    bad_act = np.array([[0, 0.5, 11, 3.0]])
    with pytest.raises(IndexError):
        dc.RecordTable.from_reference_table(bad_act, 5, 11)
    with pytest.raises(ValueError):
        dc.RecordTable.from_reference_table(np.zeros((3, 3)), 1, 11)


def test_refused_inputs_raise_where_the_reference_wraps_or_picks_nan(dc, golden):
    """tests/golden/refused_inputs.npz records what the UNMODIFIED reference does with a negative state id (S2:77-80: Python indexing
    wraps it: -1 is state 19) and with a NaN reward (S2:92: np.argmax sticks to the first NaN while max() skips it) —
    tests/test_oracle_golden.py replays both.  This library reproduces neither: the same tables raise at its boundary, through every
    builder, and the clean part of each table still runs."""
    g = golden("refused_inputs.npz")
    neg, nan = g["negative_id_data"], g["nan_reward_data"]
    for build in (lambda d: dc.RecordTable.from_reference_table(d, 20, 11),
                  lambda d: dc.RecordTable.from_reference_table(d, 20, 11, storage=torch.float64),
                  lambda d: dc.RecordTable.from_reference_table(d, 20, 11, arrival=False),
                  lambda d: dc.records.buckets_from_reference_table(d, 20, 11),
                  lambda d: dc.reference_api.run_simulation(d, g["action_value"], 20, 11)):
        with pytest.raises(IndexError):
            build(neg)
        with pytest.raises(ValueError):
            build(nan)
    with pytest.raises(ValueError):
        dc.ConfidenceEstimator().bounds(torch.from_numpy(nan[:, 3].astype(np.float32)).cuda(), 1, 1, n_dense=len(nan), check_finite=True)
    ok = neg[neg[:, 0] >= 0]                                                # the rows the reference files where they say
    tr = dc.ConfidenceEstimator().trace(dc.RecordTable.from_reference_table(ok, 20, 11, storage=torch.float64))
    want = g["negative_id_bucket_len"].copy()
    want[19] = 0
    want[0] -= np.bincount(neg[neg[:, 0] == -20, 2].astype(int), minlength=11)
    assert np.array_equal(tr.n.cpu().numpy(), want)


# ---- scan -------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N", [1, 2047, 2048, 2049, 1_000_003])
def test_scan(dc, N):
    from dcarl_amd import _lib
    lib = dc.load_library()
    dev = dc.require_gpu()
    x = torch.randn(N, dtype=torch.float64, device=dev)
    out = torch.empty_like(x)
    ws = torch.empty(int(lib.dcarl_scan_workspace_bytes(N)), dtype=torch.uint8, device=dev)
    _lib.check(lib.dcarl_scan_f64(_lib.ptr(x), _lib.ptr(out), N, _lib.ptr(ws), _lib.stream_ptr()))
    assert (out - torch.cumsum(x, 0)).abs().max().item() < 1e-9


# ---- BASELINE.json sizes through size-independent properties ---------------------------------------------
def test_full_size_replicas_properties(dc, golden, sim1_data):
    """configs[1] shape (Sim1 x 65 536 replicas x 20 000 records, fp32) — scaled to what fits the GPU's free memory.
    Properties: (1) replica 0 carries the real bundled samples and must reproduce the Sim1 golden exactly;
    (2) replicas built from identical streams give identical traces; (3) final step value == vmax, bucket sizes
    sum to the stream length; (4) the online kernel's final table == the batch kernel's on the same samples."""
    dev = dc.require_gpu()
    free, _ = torch.cuda.mem_get_info()
    T = 20000
    S = 65536
    while S * T * 10 * 1.3 > free * 0.8 and S > 1024:
        S //= 2
    q = torch.from_numpy(sim1_data[1].astype(np.float32))          # (1,11) shared by all replicas
    tbl = dc.sampler.sample_state_records(q, T, seed=0, S=S)
    d = sim1_data[0][:T]
    t = torch.arange(T, device=dev)
    e0 = tbl.elem(torch.zeros(T, dtype=torch.int64, device=dev), t)
    e1 = tbl.elem(torch.full((T,), 1, dtype=torch.int64, device=dev), t)
    e2 = tbl.elem(torch.full((T,), S - 1, dtype=torch.int64, device=dev), t)
    tbl.R[e0] = torch.from_numpy(d[:, 3].astype(np.float32)).to(dev)
    tbl.act[e0] = torch.from_numpy(d[:, 2].astype(np.uint8)).to(dev)
    tbl.R[e2] = tbl.R[e1]
    tbl.act[e2] = tbl.act[e1]
    tr = dc.ConfidenceEstimator().trace(tbl)
    g = golden("sim1_trace.npz")
    assert np.array_equal(tr.step_act[e0].cpu().numpy(), g["step_act"])
    assert rel(tr.step_val[e0].double().cpu().numpy(), g["step_value"]).max() <= 1e-5
    assert int(tr.activation_step[0]) == 4438
    assert torch.equal(tr.step_val[e1], tr.step_val[e2]) and torch.equal(tr.step_act[e1], tr.step_act[e2])
    last = tbl.elem(torch.arange(S, device=dev), torch.full((S,), T - 1, device=dev))
    assert torch.equal(tr.step_val[last], tr.vmax)
    assert torch.equal(tr.step_act[last].to(torch.int32), tr.amax)
    assert torch.equal(tr.n.sum(1), torch.full((S,), T, dtype=torch.int64, device=dev))
    # activation latch is consistent with the step trace: first t with arg-max != 0
    sub = torch.arange(0, S, max(1, S // 512), device=dev)
    tt = torch.arange(T, device=dev)
    ee = tbl.elem(sub[:, None].expand(-1, T).reshape(-1), tt[None].expand(len(sub), -1).reshape(-1)).view(len(sub), T)
    nz = tr.step_act[ee] != 0
    first = torch.where(nz.any(1), nz.float().argmax(1) + 1, torch.full((len(sub),), -1, device=dev))
    assert torch.equal(first.to(torch.int32), tr.activation_step[sub])
    # batch kernel on the same samples (sorted by action inside each sampled state)
    sub_cpu = sub[:64]
    ee = ee[:64]
    R = tbl.R[ee].cpu().numpy()
    a = tbl.act[ee].cpu().numpy()
    vals, seg = [], [0]
    for i in range(len(sub_cpu)):
        o = np.argsort(a[i], kind="stable")
        vals.append(R[i][o])
        seg.extend((seg[-1] + np.cumsum(np.bincount(a[i], minlength=11))).tolist())
    res = dc.ConfidenceEstimator().bounds(torch.from_numpy(np.concatenate(vals)).to(dev), len(sub_cpu), 11,
                                          seg_off=torch.tensor(seg, dtype=torch.int64))
    assert rel(res.V.cpu().numpy(), tr.V[sub_cpu].cpu().numpy()).max() <= 1e-9
    assert torch.equal(res.amax, tr.amax[sub_cpu])


# ---- sampler entry points in distribution -------------------------------------------------------
def test_random_state_manual_chi2(dc):
    """DS:19-28: 10 % of the draws are state 0, the others uniform on 1..S-1."""
    from scipy import stats
    api = dc.reference_api
    api.seed(2024)
    S, N = 20, 40000
    m = np.asarray(api.random_state_manual(S, N, legacy_streams=False))
    assert m.min() >= 0 and m.max() <= S - 1
    obs = np.bincount(m, minlength=S)
    exp = np.array([0.1] + [0.9 / (S - 1)] * (S - 1)) * N
    assert stats.chisquare(obs, exp).pvalue > 1e-3
    assert abs(obs[0] / N - 0.1) < 0.006


def test_random_state_norm_chi2_vs_reference_histogram(dc, golden):
    """DS:12-17 through the library's own Philox stream: the histogram of floor(N(3,1)/6*20) against (i) the law and
    (ii) the reference's own draws stored in the goldens (3 seeds x 1000 draws), as a two-sample chi-square."""
    from scipy import stats
    api = dc.reference_api
    api.seed(7)
    r = api.random_state_norm(20, 200000, legacy_streams=False)
    assert r.dtype.kind == "i"
    edges = np.arange(-4, 26)                                   # values outside [0,20) are legal (DS:50-51 filters later)
    obs = np.histogram(r, bins=edges)[0]
    p = np.diff(stats.norm.cdf(6.0 * edges / 20.0 - 3.0))
    keep = p * len(r) >= 5
    chi = ((obs[keep] - p[keep] * len(r)) ** 2 / (p[keep] * len(r))).sum()
    assert stats.chi2.sf(chi, keep.sum() - 1) > 1e-3
    ref = np.concatenate([golden(f"sampler_seed{s}.npz")["random_state_norm_out"] for s in (0, 1, 2)])
    ref_h = np.histogram(ref, bins=edges)[0]
    both = (obs + ref_h) > 0
    rows = np.stack([obs[both], ref_h[both]])
    merged = rows[:, rows[1] >= 5]                              # pool the thin tails of the 3000-draw reference sample
    tail = rows[:, rows[1] < 5].sum(1, keepdims=True)
    table = np.concatenate([merged, tail], 1) if tail.sum() else merged
    assert stats.chi2_contingency(table)[1] > 1e-3


def test_add_an_act_data_ks(dc):
    """DS:5-9: R ~ N(Q[act], 50).  4 000 scalar calls of the drop-in function + 1e5 draws through the kernel it calls."""
    from scipy import stats
    api = dc.reference_api
    api.seed(99)
    q = np.linspace(-50, 100, 11)
    x = np.array([api.add_an_act_data(i % 11, q, legacy_streams=False) for i in range(4000)])
    z = (x - q[np.arange(4000) % 11]) / 50.0
    assert stats.kstest(z, "norm").pvalue > 1e-3
    tbl = dc.sampler.sample_state_records(torch.tensor([[25.0]]), 100000, seed=5, stream_id=9)
    big = tbl.R[tbl.state_major_index()].double().cpu().numpy()
    assert stats.kstest((big - 25.0) / 50.0, "norm").pvalue > 1e-3
    assert abs(big.mean() - 25.0) < 0.6 and abs(big.std() - 50.0) < 0.5


def test_narrowed_launch_equals_the_full_one(dc):
    """A > 16 with never-sampled trailing candidates (the Sim1 script: action_num = 30, 11 sampled): the estimator runs the
    loop on candidates 0 .. max_action + 1 and pads the table.  Same trace as the 32-slot kernel and as the oracle —
    including states where every sampled candidate (the rule action too) has sunk below the -50 prior, so that a
    NEVER-sampled candidate is the arg-max: the first one, which may lie inside or just above the sampled range."""
    rng = np.random.RandomState(21)
    S, A, T = 140, 30, 400
    used = np.array([0, 1, 2, 4, 5, 7, 9])                      # 3, 6, 8 are never sampled either; 10.. are the tail
    act = used[rng.randint(0, len(used), S * T)].astype(np.uint8)
    q = rng.uniform(-50, 100, (S, A))
    q[:40] = -500.0                                             # states 0..39: everything sampled ends far below -50
    st = np.repeat(np.arange(S), T)
    R = q[st, act] + 50 * rng.standard_normal(S * T)
    est = dc.ConfidenceEstimator()
    tbl = dc.RecordTable.from_state_major(R, act, np.full(S, T), A, storage=torch.float64)
    assert tbl.max_action == 9
    tr = est.trace(tbl)
    assert dc._lib.last_kernel().startswith("trace_nwave_kernel<double,11,")          # 0 .. 10: ten sampled ids + 1 stand-in
    tbl.max_action = None
    full = est.trace(tbl)
    assert dc._lib.last_kernel().startswith("trace_kernel<double,32>")
    assert torch.equal(tr.step_act, full.step_act) and torch.equal(tr.step_val, full.step_val)
    assert torch.equal(tr.V, full.V) and torch.equal(tr.n, full.n) and torch.equal(tr.activation_step, full.activation_step)
    assert torch.equal(tr.amax, full.amax) and torch.equal(tr.vmax, full.vmax)
    ref = co.trace(R, act, np.arange(S + 1, dtype=np.int64) * T, S, A)
    assert np.array_equal(tr.steps_by_state()[1].cpu().numpy(), ref["step_act"])
    assert np.array_equal(tr.amax.cpu().numpy(), ref["amax"]) and set(ref["amax"][:40]) == {3}   # first never-sampled id
    assert np.array_equal(tr.V.cpu().numpy()[:, 10:], np.full((S, 20), -50.0)) and not tr.n.cpu().numpy()[:, 10:].any()
    # never-sampled candidates only above the range: the stand-in itself wins
    act2 = rng.randint(0, 5, S * T).astype(np.uint8)
    R2 = -500.0 + 50 * rng.standard_normal(S * T)
    t2 = dc.RecordTable.from_state_major(R2, act2, np.full(S, T), 24, storage=torch.float32)
    r2 = est.trace(t2)
    assert dc._lib.last_kernel().startswith("trace_nwave_kernel<float,6,")
    ref2 = co.trace(R2.astype(np.float32), act2, np.arange(S + 1, dtype=np.int64) * T, S, 24)
    assert np.array_equal(r2.amax.cpu().numpy(), ref2["amax"]) and set(ref2["amax"]) == {5}
    assert np.array_equal(r2.steps_by_state()[1].cpu().numpy(), ref2["step_act"])
    # reuse of the output buffers (bench.py's timed loop)
    r3 = est.trace(t2, out=r2)
    assert torch.equal(r3.V, r2.V) and r3.V.shape == (S, 24)


def test_narrowed_launch_keeps_a_never_sampled_non_rule_candidate(dc):
    """rule_act == max_action + 1 with swapped priors (init_other > init_rule).  The last kept candidate must not
    be the rule action itself: a never-sampled NON-rule candidate at init_other has to stay in the launch, because it wins
    as soon as every sampled value (and the rule prior) lies below it."""
    rng = np.random.RandomState(5)
    S, A, T = 70, 30, 300
    act = rng.randint(0, 6, S * T).astype(np.uint8)                 # ids 0..5 sampled; rule action 6 = max_action + 1
    R = -400.0 + 50 * rng.standard_normal(S * T)
    p = dc.Params(rule_act=6, init_rule=-80.0, init_other=-20.0)
    est = dc.ConfidenceEstimator(p)
    tbl = dc.RecordTable.from_state_major(R, act, np.full(S, T), A, storage=torch.float64)
    assert tbl.max_action == 5
    tr = est.trace(tbl)
    assert dc._lib.last_kernel().startswith("trace_nwave_kernel<double,8,")            # 0..5, the rule action 6, stand-in 7
    tbl.max_action = None
    full = est.trace(tbl)
    assert dc._lib.last_kernel().startswith("trace_kernel<double,32>")
    for k in ("step_act", "step_val", "V", "n", "amax", "vmax", "activation_step"):
        assert torch.equal(getattr(tr, k), getattr(full, k)), k
    ref = co.trace(R, act, np.arange(S + 1, dtype=np.int64) * T, S, A,
                   co.params(rule_act=6, init_rule=-80.0, init_other=-20.0))
    assert np.array_equal(tr.steps_by_state()[1].cpu().numpy(), ref["step_act"])
    assert np.array_equal(tr.amax.cpu().numpy(), ref["amax"]) and set(ref["amax"]) == {7}


def test_trace_out_reuse_across_narrowings(dc):
    """a TraceResult reused for a table with another narrowing re-allocates the narrow buffers and re-initialises
    the padded columns; one made for another shape is refused."""
    rng = np.random.RandomState(9)
    S, A, T = 80, 28, 200
    est = dc.ConfidenceEstimator()

    def table(hi):
        act = rng.randint(0, hi, S * T).astype(np.uint8)
        R = 20.0 + 50 * rng.standard_normal(S * T)
        return dc.RecordTable.from_state_major(R, act, np.full(S, T), A, storage=torch.float32)
    wide, narrow = table(20), table(7)                              # narrowed to 21 candidates vs 8
    fresh_w, fresh_n = est.trace(wide), est.trace(narrow)
    out = est.trace(wide)
    again = est.trace(narrow, out=out)                              # out came from another a_run
    for k in ("step_act", "step_val", "V", "n", "amax", "activation_step"):
        assert torch.equal(getattr(again, k), getattr(fresh_n, k)), k
    back = est.trace(wide, out=again)
    for k in ("step_act", "V", "n", "amax"):
        assert torch.equal(getattr(back, k), getattr(fresh_w, k)), k
    small = dc.RecordTable.from_state_major(np.zeros(40), np.zeros(40, dtype=np.uint8), [40], A)
    with pytest.raises(ValueError):
        est.trace(small, out=back)


# ---- a12-a15 seed-compatible: the reference's own random sources, the arithmetic on the GPU --------------
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_state_manual_injected_streams_bit_exact(dc, golden, seed):
    """a14 (DS:19-28): the kernel on the captured random.random() / random.randint streams == the reference's output, and the
    drop-in function after random.seed(...) == the reference's output."""
    import random
    g = golden(f"sampler_seed{seed}.npz")
    out = dc.sampler.state_manual_from_streams(g["random_state_manual_u"], g["random_state_manual_r"])
    assert np.array_equal(out.cpu().numpy(), g["random_state_manual_out"])
    assert np.array_equal(np.asarray(orc.random_state_manual_from_streams(g["random_state_manual_u"], g["random_state_manual_r"])),
                          g["random_state_manual_out"])
    random.seed(seed + 200)                                       # how tests/golden/make_goldens.py seeded the reference
    got = dc.reference_api.random_state_manual(20, 1000)
    assert isinstance(got, list) and got == g["random_state_manual_out"].tolist()
    assert dc.reference_api.random_state_manual(20, 0) == []
    # all-zero and no-zero streams
    assert dc.sampler.state_manual_from_streams([0.05, 0.1, 0.0], []).tolist() == [0, 0, 0]      # 0.1 itself is NOT > 0.1
    assert dc.sampler.state_manual_from_streams([0.5, 0.11], [7, 3]).tolist() == [7, 3]
    with pytest.raises(ValueError):
        dc.sampler.state_manual_from_streams([0.5, 0.6], [1])


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_seeded_drop_in_sampler_equals_the_reference_bit_for_bit(dc, golden, seed, tmp_path, monkeypatch):
    """np.random.seed(s); random.seed(s); Data_Generation() — seeded the way one would seed the reference — writes the
    reference's three files bit for bit (legacy streams: NumPy's global RandomState + Python's random, arithmetic on the GPU)."""
    import random
    api = dc.reference_api
    g = golden(f"sampler_seed{seed}.npz")
    monkeypatch.chdir(tmp_path)
    (tmp_path / "Simulation_testing" / "Simulation_Data_Collection").mkdir(parents=True)
    np.random.seed(seed); random.seed(seed)
    assert api.Data_Generation() is None
    base = "Simulation_testing/Simulation_Data_Collection/"
    data, q, states = np.load(base + "data.npy"), np.load(base + "action_value.npy"), np.load(base + "states.npy")
    assert data.dtype == np.float64 and np.array_equal(data, g["data"])
    assert np.array_equal(q, g["action_value"]) and np.array_equal(states, g["states"])
    np.random.seed(seed + 100)
    r = api.random_state_norm(20, 1000)
    assert r.dtype == np.asarray(g["random_state_norm_out"]).astype(int).dtype and np.array_equal(r, g["random_state_norm_out"])
    # add_an_act_data: norm.rvs(loc=Q[act], scale=50, size=1) == Q[act] + 50 * (one legacy gauss draw)
    np.random.seed(seed + 7)
    z = np.random.RandomState(seed + 7).standard_normal(3)
    got = [api.add_an_act_data(a, q[4]) for a in (2, 9, 0)]
    assert got == [float(q[4][a] + 50 * zz) for a, zz in zip((2, 9, 0), z)]


# ---- the final table without the loop's per-record work (csrc/trace_final.hip) ----------------------------------------------
@pytest.mark.parametrize("storage", [torch.float32, torch.float64])
@pytest.mark.parametrize("S,A,T,kind", [(1, 30, 20000, "uniform"), (20, 11, 2500, "ragged"), (64, 1, 50, "uniform"), (65, 5, 333, "holes"),
                                        (1000, 11, 700, "ragged"), (4096, 12, 300, "sorted"), (3000, 16, 257, "ragged"),
                                        (500, 17, 129, "ragged"), (300, 24, 64, "holes"), (130, 32, 45, "ragged"), (70000, 11, 40, "ragged")])
def test_final_table_kernel_equals_the_online_kernel(dc, S, A, T, kind, storage, knob):
    """dcarl_trace_* with no per-record output and no latch requested runs final_table_kernel: V, n, max, arg-max must equal the
    online kernel's (the loop's table after its last record, S1:86-95) bit for bit — ragged tables, sorted slots, empty states,
    every kernel family of the candidate count (multi-wave <= 16, one-wave 24 / 32), both storage types."""
    rng = np.random.RandomState(S * 31 + A)
    if kind == "uniform":
        lens = np.full(S, T)
    elif kind == "ragged":
        lens = rng.randint(0, T + 1, S)
    elif kind == "sorted":
        lens = np.sort(rng.randint(max(T - 40, 0), T + 1, S))[::-1].copy()
    else:
        lens = np.where(rng.rand(S) < 0.2, 0, T)
    n = int(lens.sum())
    R = (rng.randn(n) * 50 + rng.uniform(-50, 100)).astype(np.float32)
    R[rng.rand(n) < 0.02] = 7.25                                  # repeated values: ties between candidates do occur
    act = rng.randint(0, A, n)
    t = dc.RecordTable.from_state_major(R, act, lens, A, storage=storage)
    est = dc.ConfidenceEstimator()
    full = est.trace(t, want_steps=False).check()                 # act_step asked for: the online kernel
    assert "final_table" not in dc._lib.last_kernel()
    fin = est.trace(t, want_steps=False, want_latch=False)
    assert dc._lib.last_kernel().startswith("final_table_kernel")
    assert fin.activation_step is None
    assert torch.equal(fin.V, full.V) and torch.equal(fin.n, full.n)
    assert torch.equal(fin.amax, full.amax) and torch.equal(fin.vmax, full.vmax)
    b = est.bounds_from_table(t)
    assert torch.equal(b.V, full.V) and torch.equal(b.amax, full.amax)
    knob("DCARL_FINAL_TABLE", "0")                                # the switch back to the online kernel (A/B variant of the library)
    again = dc.ConfidenceEstimator().trace(t, want_steps=False, want_latch=False)
    assert "final_table" not in dc._lib.last_kernel()
    assert torch.equal(again.V, full.V) and torch.equal(again.amax, full.amax)
    with pytest.raises(ValueError):
        est.trace(t, want_steps=True, want_latch=False)


def test_final_table_kernel_on_the_reference_tables(dc, golden, sim1_data, sim2_data):
    """... and on the bundled tables it gives the reference's own final TSRL_value / bucket sizes (goldens)."""
    est = dc.ConfidenceEstimator()
    for which, data, S, A in (("sim1", sim1_data[0], 1, 30), ("sim2", sim2_data[0], 20, 11)):
        g = golden(f"{which}_trace.npz")
        b = est.bounds_from_table(dc.RecordTable.from_reference_table(data, S, A, storage=torch.float64, limit=20000))
        assert dc._lib.last_kernel().startswith("final_table_kernel")
        assert np.array_equal(b.n.cpu().numpy(), g["bucket_len"])
        ref = g["TSRL_value"].reshape(S, A)
        assert np.abs(b.V.cpu().numpy() - ref).max() <= 1e-9 * np.abs(ref).max()
