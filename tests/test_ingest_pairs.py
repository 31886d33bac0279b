"""The sampler's {idx, act, R} arrays straight into the online layout (dcarl_ingest_group_pairs_f32, ABI 7;
RecordTable.from_pairs): data_sampling.py's output (DS:45-55) becomes test_DCARL.py's input (S1:73-80) without the (N,4) float64
table in between.  The table must equal — bit for bit — the one from_reference_table builds of the rows DS:55 would have
appended, which in turn is pinned on a stable NumPy sort and on the reference's goldens (tests/test_ingest.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dc():
    import dcarl_amd
    dcarl_amd.require_gpu()
    return dcarl_amd


def rows_of(idx, act, R):
    """DS:50-55: the visits inside [0, state_num) as {state idx, state feature, action, cumulative reward} float64 rows."""
    keep = idx != -1
    rows = np.zeros((int(keep.sum()), 4), dtype=np.float64)
    rows[:, 0], rows[:, 2], rows[:, 3] = idx[keep], act[keep], R[keep].astype(np.float64)
    return rows


def same_table(a, b):
    assert a.S == b.S and a.A == b.A and a.n_records == b.n_records
    assert torch.equal(a.lengths, b.lengths) and torch.equal(a.slice_row_off, b.slice_row_off)
    assert (a.slot_state is None) == (b.slot_state is None)
    if a.slot_state is not None:
        assert torch.equal(a.slot_state, b.slot_state) and torch.equal(a.state_slot, b.state_slot)
    assert torch.equal(a.R, b.R) and torch.equal(a.act, b.act)
    assert a.max_action == b.max_action


@pytest.mark.parametrize("S,A,N,sort", [(20, 11, 1_000_000, True), (20, 11, 4099, False), (4096, 11, 1 << 22, True), (65536, 16, (1 << 21) + 777, True),
                                        (300, 5, 6656, True), (300, 5, 6657, False), (1, 2, 100, True), (65, 32, 50_000, True)])
def test_pairs_table_equals_the_table_of_the_rows(dc, S, A, N, sort):
    q = torch.from_numpy(np.random.RandomState(S + A).uniform(-50, 100, (S, A)).astype(np.float32))
    idx, act, R = dc.sampler.sample_pairs(q, N, seed=3, offset=5)
    t = dc.RecordTable.from_pairs(idx, act, R, S, A, sort_by_length=sort)
    h_idx, h_act, h_R = idx.cpu().numpy(), act.cpu().numpy(), R.cpu().numpy()
    dropped = int((h_idx == -1).sum())
    assert t.n_records == N - dropped and int(t.lengths.sum()) == N - dropped
    if S > 1:
        assert dropped > 0                      # the visit law floor(N(3,1)/6*S) leaves [0,S) for ~0.27 % of the draws (DS:50-51)
    ref = dc.RecordTable.from_reference_table(rows_of(h_idx, h_act, h_R), S, A, storage=torch.float32, sort_by_length=sort, arrival=False)
    same_table(t, ref)
    # independently of the row ingest: every state's records in arrival order (a stable NumPy sort of the kept pairs)
    keep = h_idx != -1
    order = np.argsort(h_idx[keep], kind="stable")
    sm = t.state_major_index()
    assert np.array_equal(t.R[sm].cpu().numpy(), h_R[keep][order])
    assert np.array_equal(t.act[sm].cpu().numpy(), h_act[keep][order].astype(np.uint8))


def test_pairs_to_estimator_equals_rows_to_estimator(dc):
    """sampler -> from_pairs -> online loop == sampler -> (N,4) rows -> from_reference_table -> online loop, every output."""
    S, A, N = 2048, 11, 1 << 21
    q = dc.workloads.uniform_q(torch.arange(S), A, seed=1)
    idx, act, R = dc.sampler.sample_pairs(q, N, seed=11)
    est = dc.ConfidenceEstimator()
    a = est.trace(dc.RecordTable.from_pairs(idx, act, R, S, A)).check()
    rows = rows_of(idx.cpu().numpy(), act.cpu().numpy(), R.cpu().numpy())
    b = est.trace(dc.RecordTable.from_reference_table(rows, S, A, storage=torch.float32, arrival=False)).check()
    for x, y in ((a.V, b.V), (a.n, b.n), (a.activation_step, b.activation_step), (a.amax, b.amax), (a.vmax, b.vmax),
                 (a.step_val, b.step_val), (a.step_act, b.step_act)):
        assert torch.equal(x, y)


def test_pairs_edge_cases_and_errors(dc):
    S, A = 300, 5
    i32 = lambda v: torch.tensor(v, dtype=torch.int32)          # noqa: E731
    f32 = lambda v: torch.tensor(v, dtype=torch.float32)        # noqa: E731
    # every visit dropped / no visit at all: an empty table
    for idx in ([-1] * 1000, []):
        t = dc.RecordTable.from_pairs(i32(idx), i32([0] * len(idx)), f32([1.0] * len(idx)), S, A)
        assert t.n_records == 0 and int(t.lengths.sum()) == 0
    # one kept record among dropped ones, in the last place of a tile and the first of the next
    for pos in (6655, 6656):
        idx = np.full(13312, -1, dtype=np.int32)
        idx[pos] = 299
        t = dc.RecordTable.from_pairs(idx, np.full(13312, 4, dtype=np.int32), np.full(13312, 2.5, dtype=np.float32), S, A)
        assert t.n_records == 1 and int(t.lengths_by_state[299]) == 1
        e = t.elem(torch.tensor(299), torch.tensor(0, device="cuda"))
        assert float(t.R[e]) == 2.5 and int(t.act[e]) == 4
    # ids past the table raise like the reference's S1:80 (negative ids other than the sampler's -1 would wrap there: refused)
    with pytest.raises(IndexError):
        dc.RecordTable.from_pairs(i32([0, 300]), i32([0, 0]), f32([1, 1]), S, A)
    with pytest.raises(IndexError):
        dc.RecordTable.from_pairs(i32([0, -2]), i32([0, 0]), f32([1, 1]), S, A)
    with pytest.raises(IndexError):
        dc.RecordTable.from_pairs(i32([0, 1]), i32([0, 5]), f32([1, 1]), S, A)
    with pytest.raises(IndexError):
        dc.RecordTable.from_pairs(i32([0, 1]), i32([-1, 0]), f32([1, 1]), S, A)
    for bad in (float("nan"), float("inf"), -float("inf")):
        with pytest.raises(ValueError):
            dc.RecordTable.from_pairs(i32([0, 1]), i32([0, 0]), f32([1, bad]), S, A)
    # a dropped visit's action and reward are never looked at
    t = dc.RecordTable.from_pairs(i32([-1, 7]), i32([99, 1]), f32([float("nan"), 3.0]), S, A)
    assert t.n_records == 1
    with pytest.raises(ValueError):
        dc.RecordTable.from_pairs(i32([0, 1]), i32([0]), f32([1, 1]), S, A)
    # more states than the direct ingest serves: built through the rows, same semantics
    big = dc.RecordTable.from_pairs(i32([70000, -1, 3, 70000]), i32([1, 0, 2, 0]), f32([1.5, 9, 2.5, 3.5]), 70001, A)
    assert big.n_records == 3 and int(big.lengths_by_state[70000]) == 2
    assert float(big.R[big.elem(torch.tensor(70000), torch.tensor(1, device="cuda"))]) == 3.5


def test_pairs_entry_point_refuses_what_it_does_not_serve(dc):
    from dcarl_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda")
    N, S, A = 1000, 300, 5
    idx = torch.zeros(N, dtype=torch.int32, device=dev)
    R = torch.zeros(N, dtype=torch.float32, device=dev)
    ws = torch.empty(int(lib.dcarl_ingest_workspace_bytes(N, S, A, 4, 8 | 1, 0)), dtype=torch.uint8, device=dev)
    outs = [torch.empty(S, dtype=torch.int32, device=dev) for _ in range(3)]
    sro = torch.empty(S // 64 + 2, dtype=torch.int64, device=dev)
    info = torch.empty(16, dtype=torch.int64, device=dev)

    def call(n=N, s=S, flags=8 | 1, i=idx):
        return lib.dcarl_ingest_group_pairs_f32(_lib.ptr(i), _lib.ptr(idx), _lib.ptr(R), n, s, A, flags, _lib.ptr(ws), *[_lib.ptr(o) for o in outs],
                                                _lib.ptr(sro), _lib.ptr(info), _lib.stream_ptr())
    assert call() == 0
    assert call(flags=1) == -1                 # without DCARL_INGEST_FORCE_DIRECT
    assert call(flags=8 | 2) == -1             # arrival bookkeeping is the sort path's
    assert call(flags=8 | 4) == -1
    assert call(n=0) == -1
    assert call(s=65537) == -1
    assert call(i=None) == -1
    assert b"dcarl_ingest_group_pairs" in lib.dcarl_last_error()
    torch.cuda.synchronize()


# ---- host-compacted records (ABI 8: dcarl_host_compact_rows_f32 -> dcarl_ingest_group_packed_f32; RecordTable.from_packed) -------------
@pytest.mark.parametrize("S,A,N,sort", [(20, 11, 1_000_000, True), (20, 11, 4099, False), (4096, 11, 1 << 22, True), (65536, 16, (1 << 21) + 777, True),
                                        (300, 5, 6656, True), (300, 5, 6657, False), (1, 2, 100, True), (65, 32, 50_000, True), (7, 3, 1, True)])
def test_packed_table_equals_the_table_of_the_rows(dc, S, A, N, sort):
    """(N,4) float64 rows -> 8-byte records on the HOST -> the direct ingest on the device == from_reference_table of the rows, bit for
    bit (layout, lengths, slot order, every record), with fractional ids (truncated toward zero like int(), S1:77-78) and rewards that
    are not f32-representable (rounded once, on the host, exactly as the device would)."""
    from dcarl_amd.records import compact_rows_host
    rng = np.random.default_rng(S * 31 + A)
    rows = np.empty((N, 4), dtype=np.float64)
    rows[:, 0] = rng.integers(0, S, N) + rng.random(N) * 0.99
    rows[:, 1] = rng.random(N)
    rows[:, 2] = rng.integers(0, A, N) + rng.random(N) * 0.99
    rows[:, 3] = rng.normal(20.0, 50.0, N)
    rec, info = compact_rows_host(rows, S, A)
    assert info[8] == N and info[7] == 0
    t = dc.RecordTable.from_packed(torch.from_numpy(rec), S, A, sort_by_length=sort)
    ref = dc.RecordTable.from_reference_table(rows, S, A, storage=torch.float32, sort_by_length=sort, arrival=False)
    same_table(t, ref)


def test_packed_records_are_rechecked_on_the_device(dc):
    """A packed record whose ids do not fit the table (a caller that skipped the host-side info, or a corrupt buffer) is filed under id 0
    and reported: IndexError / ValueError from the same info words as any other ingest."""
    S, A = 300, 5
    def pack(st, a, r):
        return (np.uint64(st) << np.uint64(5)) | np.uint64(a) | (np.asarray(r, np.float32).view(np.uint32).astype(np.uint64) << np.uint64(32))
    good = pack(np.arange(200) % S, np.arange(200) % A, np.linspace(-5, 5, 200))
    t = dc.RecordTable.from_packed(torch.from_numpy(good.view(np.int64)), S, A)
    assert t.n_records == 200
    bad = good.copy()
    bad[17] = pack(S + 3, 1, 0.0)
    with pytest.raises(IndexError):
        dc.RecordTable.from_packed(torch.from_numpy(bad.view(np.int64)), S, A)
    bad = good.copy()
    bad[5] = pack(2, A, 0.0)
    with pytest.raises(IndexError):
        dc.RecordTable.from_packed(torch.from_numpy(bad.view(np.int64)), S, A)
    bad = good.copy()
    bad[9] = pack(2, 1, np.inf)
    with pytest.raises(ValueError):
        dc.RecordTable.from_packed(torch.from_numpy(bad.view(np.int64)), S, A)
    with pytest.raises(ValueError):
        dc.RecordTable.from_packed(torch.zeros(0, dtype=torch.int64), S, A)
    with pytest.raises(ValueError):
        dc.RecordTable.from_packed(torch.zeros(4, dtype=torch.int32), S, A)
