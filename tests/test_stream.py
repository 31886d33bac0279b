"""The online loop over a HOST-resident record table fed chunk by chunk (dcarl_amd/stream.py; ABI 7: dcarl_host_pin /
dcarl_copy_h2d / _d2h).

The reference loads ``data.npy`` into host memory and walks it front to back (S1:33,73; S2:32,72).  The streamed pipeline
(copy of chunk k+1 under the ingest + online kernel of chunk k, two HIP streams) must give bit for bit what one pass over
the whole table on the device gives — and, on the bundled tables, the reference's own numbers (golden fixtures)."""
import os
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dc():
    import dcarl_amd
    dcarl_amd.require_gpu()
    return dcarl_amd


def one_pass(dc, data, S, A, storage, with_overall=False):
    est = dc.ConfidenceEstimator()
    tr = est.trace(dc.RecordTable.from_reference_table(data, S, A, storage=storage)).check()
    sv, sa = tr.steps_in_arrival_order()
    ov = est.overall_value(tr).cpu().numpy() if with_overall else None
    return tr, sv.cpu().numpy(), sa.cpu().numpy(), ov


@pytest.mark.parametrize("pin", ["register", "stage"])
@pytest.mark.parametrize("storage", [torch.float64, torch.float32])
def test_streamed_sim2_equals_one_pass_and_the_reference(dc, golden, sim2_data, storage, pin):
    from dcarl_amd.stream import trace_stream
    data = sim2_data[0][:20000]
    # pin="register" page-locks the array it is given: a PRIVATE copy that no pageable torch copy has touched (module docstring
    # of dcarl_amd.stream: the HIP runtime keeps its own record of host ranges it locked for earlier copies of an array)
    mine = data.copy()
    g = golden("sim2_trace.npz")
    tr, sv1, sa1, ov1 = one_pass(dc, data, 20, 11, storage, with_overall=True)
    for chunk in (20000, 7001, 4096, 333):
        r = trace_stream(mine, 20, 11, chunk_records=chunk, storage=storage, want_steps=True, with_overall=True, pin=pin)
        assert r.n_records == 20000 and r.chunks == -(-20000 // chunk) and r.pinned == ("registered" if pin == "register" else "staged")
        assert np.array_equal(r.step_act, sa1), chunk
        assert np.array_equal(r.step_val, sv1), chunk
        assert np.max(np.abs(r.overall_value - ov1) / np.maximum(np.abs(ov1), 1.0)) <= 1e-12, chunk     # same deltas, another summation grouping
        assert torch.equal(r.state.V, tr.V) and torch.equal(r.state.n, tr.n) and torch.equal(r.state.act_step, tr.activation_step)
    # the reference's own numbers
    assert np.array_equal(r.state.act_step.cpu().numpy(), g["activation_step"])
    assert np.array_equal(r.state.n.cpu().numpy(), g["bucket_len"])
    if storage == torch.float64:
        assert abs(r.overall_value[-1] - 597.7193818873668) < 1e-8          # S2:105, SURVEY 8(c)


def test_streamed_limit_memmap_and_iterable_sources(dc, sim2_data):
    """data[0:limit] (S1:73), an np.memmap of the .npy file (never wholly in memory) and an iterable of ragged pieces give the
    same state; the remaining rows can be appended later to the same state (S2:72 stops at 20 000 of 49 866)."""
    from dcarl_amd.stream import trace_stream
    full = sim2_data[0]
    tr_all, sv_all, sa_all, _ = one_pass(dc, full, 20, 11, torch.float64)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "Simulation_testing/Simulation_2/data.npy")
    mm = np.load(path, mmap_mode="r")
    a = trace_stream(mm, 20, 11, chunk_records=6000, storage=torch.float64, want_steps=True, limit=20000)
    assert a.pinned == "staged" and a.n_records == 20000
    assert np.array_equal(a.step_act, sa_all[:20000]) and np.array_equal(a.step_val, sv_all[:20000])
    # ... the other 29 866 rows later, as an iterable of uneven pieces (one of them empty), into the same state
    pieces = [full[20000:20001], full[20001:20001], full[20001:33333], full[33333:]]
    b = trace_stream(iter(pieces), 20, 11, chunk_records=5000, storage=torch.float64, want_steps=True, state=a.state)
    assert b.n_records == full.shape[0] - 20000
    assert np.array_equal(b.step_act, sa_all[20000:]) and np.array_equal(b.step_val, sv_all[20000:])
    assert torch.equal(b.state.V, tr_all.V) and torch.equal(b.state.n, tr_all.n) and torch.equal(b.state.act_step, tr_all.activation_step)


def test_streamed_large_table_takes_the_direct_ingest_and_equals_one_pass(dc):
    """2^22 records over 4 096 states in chunks of 2^20 (the direct ingest's threshold): the state after the stream equals the
    single pass over the device-resident table bit for bit."""
    from dcarl_amd.stream import trace_stream
    rng = np.random.default_rng(7)
    S, A, N = 4096, 11, 1 << 22
    data = np.empty((N, 4), dtype=np.float64)
    data[:, 0] = rng.integers(0, S, N)
    data[:, 1] = rng.random(N)
    data[:, 2] = rng.integers(0, A, N)
    data[:, 3] = rng.normal(20.0, 50.0, N).astype(np.float32)
    est = dc.ConfidenceEstimator()
    mine = data.copy()                                            # (registered below: a private copy, see above)
    tr = est.trace(dc.RecordTable.from_reference_table(data, S, A, storage=torch.float32, arrival=False), want_steps=False).check()
    r = trace_stream(data, S, A, chunk_records=1 << 20, storage=torch.float32, copy_threads=4)
    # the default for such a table (f32 storage, no per-record traces): the staging threads COMPACT the rows (ABI 8,
    # dcarl_host_compact_rows_f32) and 8 instead of 32 bytes per record cross the link
    assert r.chunks == 4 and r.pinned == "staged+compacted" and r.link_bytes == 8 * N
    assert torch.equal(r.state.n, tr.n) and torch.equal(r.state.act_step, tr.activation_step)
    assert torch.equal(r.state.V, tr.V)
    assert r.step_val is None and r.overall_value is None
    r0 = trace_stream(data, S, A, chunk_records=1 << 20, storage=torch.float32, copy_threads=4, compact="off")
    assert r0.pinned == "staged" and r0.link_bytes == 32 * N       # the 32-byte rows, as before: the same state bit for bit
    assert torch.equal(r0.state.V, tr.V) and torch.equal(r0.state.n, tr.n) and torch.equal(r0.state.act_step, tr.activation_step)
    # page-locked in place: usable and un-registered again afterwards — a second stream registers it anew
    for chunk in ((1 << 21) + 12345, 1 << 20):
        r2 = trace_stream(mine, S, A, chunk_records=chunk, storage=torch.float32, pin="register")
        assert r2.pinned == "registered" and torch.equal(r2.state.V, tr.V) and torch.equal(r2.state.act_step, tr.activation_step)
    assert np.array_equal(mine, data)


def test_stream_argument_errors(dc):
    from dcarl_amd.stream import trace_stream
    good = np.zeros((8, 4))
    with pytest.raises(ValueError):
        trace_stream(np.zeros((8, 3)), 1, 2)
    with pytest.raises(ValueError):
        trace_stream(good.astype(np.float32), 1, 2)
    with pytest.raises(ValueError):
        trace_stream(good, 1, 2, chunk_records=0)
    with pytest.raises(ValueError):
        trace_stream(good[::2], 1, 2, pin="register")          # a strided view cannot be registered as one range
    with pytest.raises(ValueError):
        trace_stream(good, 1, 2, pin="pageable")
    with pytest.raises(ValueError):
        trace_stream(torch.zeros((8, 4), dtype=torch.float64, device="cuda"), 1, 2)
    bad = good.copy()
    bad[3, 0] = 5                                               # state id out of range: the reference raises IndexError (S1:80)
    with pytest.raises((ValueError, IndexError)):
        trace_stream(bad, 2, 2)
    r = trace_stream(np.zeros((0, 4)), 3, 2, want_steps=True)
    assert r.n_records == 0 and r.chunks == 0 and r.step_val.size == 0
    assert torch.equal(r.state.act_step.cpu(), torch.full((3,), -1, dtype=torch.int32))


def test_integration_md_section_2b_runs_as_written(dc, golden):
    """The host-table / pairs snippet of INTEGRATION.md section 2b, executed as written."""
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(repo, "INTEGRATION.md")).read()
    at = text.index("### 2b.")
    start = text.index("```python", at) + len("```python")
    code = text[start:text.index("```", start)]
    cwd = os.getcwd()
    os.chdir(repo)
    try:
        ns = {}
        exec(compile(code, "INTEGRATION.md#2b", "exec"), ns)
    finally:
        os.chdir(cwd)
    g = golden("sim2_trace.npz")
    # (r.state IS r2.state: the second call advanced it in place through the other 29 866 rows; a latch never resets, S1:98-99)
    latched = g["activation_step"] >= 0
    assert ns["r"].state is ns["r2"].state and ns["r"].n_records == 20000
    assert np.array_equal(ns["r"].state.act_step.cpu().numpy()[latched], g["activation_step"][latched])
    assert abs(float(ns["r"].overall_value[-1]) - float(g["overall_value"][-1])) < 1e-4 * abs(float(g["overall_value"][-1]))      # (f32 storage)
    assert ns["r2"].n_records == 29866 and int(ns["r2"].state.records_seen.sum()) == 49866
    assert ns["tr"].table.n_records == int((ns["idx"] != -1).sum()) > 49000


def test_compacted_stream_small_chunks_sources_and_errors(dc, sim2_data, golden):
    """Host compaction on the bundled table (20 states: every chunk far below the direct ingest's usual threshold), cut anywhere, from an
    np.memmap and from an iterable; the same state as one pass bit for bit; and the reference's own errors raised from the HOST-side
    validation, before the offending chunk is copied (a negative id: IndexError where the reference would wrap, S2:77-80; NaN: ValueError)."""
    from dcarl_amd.stream import trace_stream
    full = sim2_data[0]
    est = dc.ConfidenceEstimator()
    tr = est.trace(dc.RecordTable.from_reference_table(full, 20, 11, storage=torch.float32, arrival=False), want_steps=False).check()
    for chunk in (len(full), 20000, 7001, 333):
        r = trace_stream(full, 20, 11, chunk_records=chunk, storage=torch.float32, compact="on")
        assert r.pinned == "staged+compacted" and r.n_records == len(full) and r.link_bytes == 8 * len(full)
        assert torch.equal(r.state.V, tr.V) and torch.equal(r.state.n, tr.n) and torch.equal(r.state.act_step, tr.activation_step), chunk
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "Simulation_testing/Simulation_2/data.npy")
    mm = np.load(path, mmap_mode="r")
    a = trace_stream(mm, 20, 11, chunk_records=9000, storage=torch.float32, limit=20000)
    b = trace_stream(iter([full[20000:20001], full[20001:33333][::1], full[33333:]]), 20, 11, chunk_records=5000, storage=torch.float32, state=a.state)
    assert a.pinned == b.pinned == "staged+compacted"
    assert torch.equal(b.state.V, tr.V) and torch.equal(b.state.n, tr.n) and torch.equal(b.state.act_step, tr.activation_step)
    # a strided view (not C-contiguous rows): compacted through a contiguous copy of the piece
    wide = np.zeros((len(full), 6))
    wide[:, :4] = full
    c = trace_stream(wide[:, :4], 20, 11, chunk_records=10000, storage=torch.float32)
    assert torch.equal(c.state.V, tr.V)
    # what is NOT compacted: f64 storage, per-record traces, registered arrays; compact="on" says so
    assert trace_stream(full[:5000], 20, 11, storage=torch.float64).pinned == "staged"
    assert trace_stream(full[:5000], 20, 11, storage=torch.float32, want_steps=True).pinned == "staged"
    with pytest.raises(ValueError):
        trace_stream(full[:5000], 20, 11, storage=torch.float64, compact="on")
    g = golden("refused_inputs.npz")
    with pytest.raises(IndexError):
        trace_stream(g["negative_id_data"], 20, 11, chunk_records=100)
    with pytest.raises(ValueError):
        trace_stream(g["nan_reward_data"], 20, 11, chunk_records=100)
    bad = full[:1000].copy()
    bad[900, 2] = 11
    with pytest.raises(IndexError):
        trace_stream(bad, 20, 11, chunk_records=256)
