"""Continuation of the online loop (dcarl_trace_resume_*, ABI version 6).

The reference's loop is incremental — ``data_state_act``, ``TSRL_value`` and ``activation_step`` live across records
(S1:41-59,73-99) and Simulation_2 stops at ``data[0:20000]`` of 49 866 rows (S2:72).  Feeding a table in k chunks through
``ConfidenceEstimator.trace(table, state=...)`` must give bit for bit what ONE pass over the whole table gives: step traces,
table, bucket sizes, arg-max, latch."""
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


def run_chunks_arrival(dc, data, S, A, cuts, storage):
    """The (N,4) reference table fed in row ranges [cuts[i], cuts[i+1]); returns the per-arrival traces and the final state."""
    est = dc.ConfidenceEstimator()
    st = est.new_state(S, A)
    sv_parts, sa_parts = [], []
    last = None
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        tbl = dc.RecordTable.from_reference_table(data[lo:hi], S, A, storage=storage)
        last = est.trace(tbl, state=st)
        if hi > lo:
            sv, sa = last.steps_in_arrival_order()
            sv_parts.append(sv)
            sa_parts.append(sa)
    last.check()
    return torch.cat(sv_parts), torch.cat(sa_parts), st, last


@pytest.mark.parametrize("which,S,A", [("sim1", 1, 30), ("sim2", 20, 11)])
@pytest.mark.parametrize("storage", [torch.float64, torch.float32])
def test_chunks_of_the_bundled_tables_equal_one_pass(dc, golden, sim1_data, sim2_data, which, S, A, storage):
    data = (sim1_data if which == "sim1" else sim2_data)[0][:20000]
    g = golden(f"{which}_trace.npz")
    est = dc.ConfidenceEstimator()
    one = est.trace(dc.RecordTable.from_reference_table(data, S, A, storage=storage))
    sv1, sa1 = one.steps_in_arrival_order()
    # cut points: a single record, mid-quad, an EMPTY chunk, just before / after Sim1's activation step (4438), uneven tails
    for cuts in ([0, 1, 7, 7, 4437, 4438, 4439, 10001, 20000], [0, 13, 20000], [0, 0, 19999, 20000], [0, 5000, 10000, 15000, 20000]):
        sv, sa, st, last = run_chunks_arrival(dc, data, S, A, cuts, storage)
        assert torch.equal(sa, sa1), cuts
        assert torch.equal(sv, sv1), cuts
        assert torch.equal(st.V, one.V) and torch.equal(st.n, one.n) and torch.equal(st.act_step, one.activation_step), cuts
        assert torch.equal(last.amax, one.amax) and torch.equal(last.vmax, one.vmax), cuts
        assert torch.equal(st.records_seen.cpu(), torch.from_numpy(g["bucket_len"]).sum(1)), cuts
    # ... and the reference's own numbers
    assert np.array_equal(st.act_step.cpu().numpy(), g["activation_step"])
    assert np.array_equal(st.n.cpu().numpy(), g["bucket_len"])


@pytest.mark.parametrize("storage", [torch.float64, torch.float32])
def test_sim2_remaining_rows_appended_to_the_20000_row_state(dc, sim2_data, storage):
    """S2:72 consumes data[0:20000]; the other 29 866 rows of the bundled table fed LATER must land where a loop over all
    49 866 rows lands (the C oracle on the whole table)."""
    data = sim2_data[0]
    S, A = 20, 11
    N = data.shape[0]
    assert N == 49866
    sv, sa, st, last = run_chunks_arrival(dc, data, S, A, [0, 20000, N], storage)
    np_dt = np.float64 if storage == torch.float64 else np.float32
    stt = data[:, 0].astype(np.int64)
    order = np.argsort(stt, kind="stable")
    off = np.concatenate([[0], np.cumsum(np.bincount(stt, minlength=S))]).astype(np.int64)
    ref = co.trace(data[order, 3].astype(np_dt), data[order, 2].astype(np.uint8), off, S, A)
    inv = np.empty(N, np.int64)
    inv[order] = np.arange(N)                      # arrival k -> position in the state-major oracle arrays
    assert np.array_equal(sa.cpu().numpy(), ref["step_act"][inv])
    tol = 1e-10 if storage == torch.float64 else 1e-6
    err = np.abs(sv.double().cpu().numpy() - ref["step_val"][inv]) / np.maximum(1.0, np.abs(ref["step_val"][inv]))
    assert err.max() <= tol
    assert np.array_equal(st.act_step.cpu().numpy(), ref["activation_step"])
    assert np.array_equal(st.n.cpu().numpy(), ref["n"])
    assert np.abs(st.V.cpu().numpy() - ref["V"]).max() <= 1e-9
    assert np.array_equal(last.amax.cpu().numpy(), ref["amax"])
    # and bit for bit against one pass of the kernel over all rows
    one = dc.ConfidenceEstimator().trace(dc.RecordTable.from_reference_table(data, S, A, storage=storage))
    sv1, sa1 = one.steps_in_arrival_order()
    assert torch.equal(sa, sa1) and torch.equal(sv, sv1) and torch.equal(st.V, one.V) and torch.equal(st.act_step, one.activation_step)


@pytest.mark.parametrize("storage", [torch.float64, torch.float32])
def test_overall_value_continues_across_chunks(dc, golden, sim2_data, storage):
    """S2:99-105's cross-state running sum is part of the loop's state too: computed chunk by chunk (the state's record count and
    current max before the chunk + the sum the previous chunk ended with) it equals the one-pass sum, the reference's golden on
    the first 20 000 rows, and the C oracle on all 49 866."""
    data = sim2_data[0]
    S, A, N = 20, 11, data.shape[0]
    g = golden("sim2_trace.npz")
    est = dc.ConfidenceEstimator()
    one_tbl = dc.RecordTable.from_reference_table(data, S, A, storage=storage)
    one = est.trace(one_tbl)
    ov_one = est.overall_value(one)
    tol = 1e-9 if storage == torch.float64 else 1e-5
    for cuts in ([0, 20000, N], [0, 1, 163, 164, 20000, 20000, 31234, N], [0, 7, N]):
        st = est.new_state(S, A)
        parts = []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            tr = est.trace(dc.RecordTable.from_reference_table(data[lo:hi], S, A, storage=storage), state=st)
            parts.append(est.overall_value(tr))
        ov = torch.cat(parts)
        assert ov.numel() == N
        err = (ov - ov_one).abs() / ov_one.abs().clamp(min=1.0)
        assert float(err.max()) <= 1e-12, cuts                                        # same deltas, another summation grouping
        ref = torch.from_numpy(g["overall_value"]).to(ov.device)
        assert float(((ov[:20000] - ref).abs() / ref.abs().clamp(min=1.0)).max()) <= tol
        assert abs(float(st.overall_total) - float(ov[-1])) == 0.0
    # the oracle on the whole table
    stt = data[:, 0].astype(np.int64)
    order = np.argsort(stt, kind="stable")
    off = np.concatenate([[0], np.cumsum(np.bincount(stt, minlength=S))]).astype(np.int64)
    np_dt = np.float64 if storage == torch.float64 else np.float32
    ref = co.trace(data[order, 3].astype(np_dt), data[order, 2].astype(np.uint8), off, S, A)
    pos = np.empty(N, np.int64)
    pos[order] = np.arange(N)
    sv = ref["step_val"].astype(np_dt).astype(np.float64)                              # the step trace in the storage type
    ov_ref = co.overall(sv, ref["activation_step"], off, stt.astype(np.int32), pos)
    assert np.abs(ov.cpu().numpy() - ov_ref).max() / np.abs(ov_ref).max() <= tol


def chunked_state_major(dc, est, R, act, lens, A, storage, k, rng, sort):
    """Every state's stream cut at k-1 random points of its own (so chunks end mid-quad, some are empty for some states and
    some states appear only in later chunks); returns per-state concatenated traces and the final state."""
    S = len(lens)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    cuts = np.sort(np.stack([rng.randint(0, l + 1, k - 1) if k > 1 else np.zeros(0, np.int64) for l in lens]), axis=1) if S else np.zeros((0, k - 1), np.int64)
    cuts = np.concatenate([np.zeros((S, 1), np.int64), cuts, lens.reshape(S, 1)], axis=1)          # (S, k+1)
    st = est.new_state(S, A)
    sv_parts = [[] for _ in range(S)]
    sa_parts = [[] for _ in range(S)]
    last = None
    for c in range(k):
        clen = cuts[:, c + 1] - cuts[:, c]
        idx = np.concatenate([np.arange(off[s] + cuts[s, c], off[s] + cuts[s, c + 1]) for s in range(S)]) if S else np.zeros(0, np.int64)
        tbl = dc.RecordTable.from_state_major(R[idx], act[idx], clen, A, storage=storage, sort_by_length=sort)
        last = est.trace(tbl, state=st)
        sv, sa = last.steps_by_state()
        sv, sa = sv.cpu(), sa.cpu()
        o = np.concatenate([[0], np.cumsum(clen)])
        for s in range(S):
            sv_parts[s].append(sv[o[s]:o[s + 1]])
            sa_parts[s].append(sa[o[s]:o[s + 1]])
    sv = torch.cat([torch.cat(p) for p in sv_parts]) if S else torch.zeros(0)
    sa = torch.cat([torch.cat(p) for p in sa_parts]) if S else torch.zeros(0, dtype=torch.uint8)
    return sv, sa, st, last


@pytest.mark.parametrize("A", [1, 5, 11, 12, 13, 16, 17, 24])
@pytest.mark.parametrize("storage", [torch.float32, torch.float64])
def test_random_chunkings_equal_one_pass_bit_for_bit(dc, A, storage):
    """Every compiled family of the online kernel (re-loaded keys up to 12 candidates, carried arg-max 13..16, the one-wave
    kernel from 17) on ragged / sorted / hole-ridden tables, 2..5 chunks."""
    rng = np.random.RandomState(1000 + A)
    est = dc.ConfidenceEstimator()
    for S, T, kind, k in ((130, 300, "ragged", 3), (64, 50, "holes", 2), (700, 90, "uniform", 4), (257, 1300, "sorted", 5), (1, 4500, "uniform", 3)):
        lens = {"uniform": np.full(S, T), "ragged": rng.randint(0, T + 1, S), "sorted": np.sort(rng.randint(max(T - 40, 0), T + 1, S))[::-1].copy(),
                "holes": np.where(rng.rand(S) < 0.2, 0, T)}[kind].astype(np.int64)
        N = int(lens.sum())
        act = rng.randint(0, A, N).astype(np.uint8)
        stt = np.repeat(np.arange(S), lens)
        R = rng.uniform(-50, 100, (S, A))[stt, act] + 50 * rng.standard_normal(N)
        R = R.astype(np.float32 if storage == torch.float32 else np.float64)
        sort = bool(rng.rand() < 0.7)
        one = est.trace(dc.RecordTable.from_state_major(R, act, lens, A, storage=storage, sort_by_length=sort))
        sv1, sa1 = one.steps_by_state()
        sv, sa, st, last = chunked_state_major(dc, est, R, act, lens, A, storage, k, rng, sort)
        tag = (A, S, T, kind, k)
        assert torch.equal(sa, sa1.cpu()), tag
        assert torch.equal(sv, sv1.cpu()), tag
        assert torch.equal(st.V, one.V) and torch.equal(st.n, one.n) and torch.equal(st.act_step, one.activation_step), tag
        assert torch.equal(last.amax, one.amax) and torch.equal(last.vmax, one.vmax), tag
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        ref = co.trace(R, act, off, S, A)
        assert np.array_equal(sa.numpy(), ref["step_act"]) and np.array_equal(st.act_step.cpu().numpy(), ref["activation_step"]), tag
    last.check()


def test_long_buckets_resume_past_the_count_root_table(dc):
    """Buckets that start a chunk beyond the 4096-entry count-root table (the resumed launch must take the compute path from
    the first record) and a chunk that crosses the table's end."""
    rng = np.random.RandomState(7)
    est = dc.ConfidenceEstimator()
    S, A, T = 64, 2, 9000
    lens = np.full(S, T, np.int64)
    act = (rng.rand(S * T) < 0.95).astype(np.uint8)                         # action 1 collects ~8 500 samples per state
    R = (np.where(act == 1, 20.0, 60.0) + 50 * rng.standard_normal(S * T)).astype(np.float32)
    one = est.trace(dc.RecordTable.from_state_major(R, act, lens, A))
    sv1, sa1 = one.steps_by_state()
    sv, sa, st, last = chunked_state_major(dc, est, R, act, lens, A, torch.float32, 4, rng, False)
    assert torch.equal(sa, sa1.cpu()) and torch.equal(sv, sv1.cpu())
    assert torch.equal(st.V, one.V) and torch.equal(st.n, one.n) and torch.equal(st.act_step, one.activation_step)


def test_resume_argument_checks(dc):
    est = dc.ConfidenceEstimator()
    tbl = dc.sampler.sample_state_records(torch.linspace(-50, 100, 11), 40, seed=3, S=100)
    with pytest.raises(ValueError):
        est.trace(tbl, state=est.new_state(99, 11))
    with pytest.raises(ValueError):
        est.trace(tbl, state=est.new_state(100, 11), out=est.trace(tbl))
    import ctypes as C
    from dcarl_amd import _lib
    lib = dc.load_library()
    st = est.new_state(100, 11)
    cs = st.c_struct()
    cs.sumsq = None
    rc = lib.dcarl_trace_resume_f32(_lib.ptr(tbl.R), _lib.ptr(tbl.act), _lib.ptr(tbl.slice_row_off), _lib.ptr(tbl.lengths), None, 100, 11,
                                    C.byref(dc.Params().to_c()), C.byref(cs), 1, None, None, None, None, _lib.stream_ptr())
    assert rc == -1 and b"state" in lib.dcarl_last_error()
