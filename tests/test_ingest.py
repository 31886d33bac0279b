"""Record ingest (csrc/ingest.hip; S1:73-80: `data_state_act[idx][act].append(R)` row by row) against a stable NumPy sort.

Bit-exact: grouping is integer / byte work, the rewards are moved (f64 storage) or rounded once to f32 (f32 storage)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dc():
    import dcarl_amd
    dcarl_amd.require_gpu()
    return dcarl_amd


def stable_seed(*parts):
    """A seed that is the same in every process (hash() of a str is randomised per interpreter: PYTHONHASHSEED), so that a failing
    table can be rebuilt from the test id."""
    import zlib
    return zlib.crc32("-".join(str(p) for p in parts).encode())


def make_table(rng, N, S, A, kind):
    d = np.empty((N, 4), dtype=np.float64)
    if kind == "uniform":
        st = rng.integers(0, S, N)
    elif kind == "skewed":                      # a few heavy states, many empty ones
        st = np.minimum((rng.exponential(S / 12.0, N)).astype(np.int64), S - 1)
    elif kind == "one_state":
        st = np.full(N, S - 1)
    elif kind == "state_major":                 # long runs of one state: every lane of a wave holds the same digit
        st = np.sort(rng.integers(0, S, N))
    elif kind == "reversed":
        st = np.sort(rng.integers(0, S, N))[::-1].copy()
    elif kind == "round_robin":
        st = np.arange(N) % S
    else:
        raise ValueError(kind)
    d[:, 0] = st + rng.random(N) * 0.9          # idx = int(idx_ori) truncates (S1:77)
    d[:, 1] = rng.random(N)
    d[:, 2] = rng.integers(0, A, N)
    d[:, 3] = rng.normal(0, 50, N)
    return d


def check_table(dc, d, S, A, storage, sort_by_length=True, arrival=True):
    tbl = dc.RecordTable.from_reference_table(d, S, A, storage=storage, sort_by_length=sort_by_length, arrival=arrival)
    N = len(d)
    st = d[:, 0].astype(np.int64)
    ac = d[:, 2].astype(np.int64)
    counts = np.bincount(st, minlength=S)
    order = np.argsort(st, kind="stable")
    assert tbl.n_records == N
    assert np.array_equal(tbl.lengths_by_state.cpu().numpy(), counts)
    assert tbl.max_action == (int(ac.max()) if N else -1)
    if sort_by_length and S > 64:
        want = np.argsort(-counts, kind="stable")
        assert np.array_equal(tbl.slot_state.cpu().numpy(), want)
        inv = np.empty(S, dtype=np.int64)
        inv[want] = np.arange(S)
        assert np.array_equal(tbl.state_slot.cpu().numpy(), inv)
        slot_len = counts[want]
    else:
        assert tbl.slot_state is None and tbl.state_slot is None
        slot_len = counts
    assert np.array_equal(tbl.lengths.cpu().numpy(), slot_len)
    # the layout's row offsets
    W = (S + 63) // 64
    pad = np.zeros(W * 64, dtype=np.int64)
    pad[:S] = slot_len
    rows = (pad.reshape(W, 64).max(1) + 3) // 4 * 4
    assert np.array_equal(tbl.slice_row_off.cpu().numpy(), np.concatenate([[0], np.cumsum(rows)]))
    # every state's records in arrival order
    idx = tbl.state_major_index()
    R_sm = tbl.R[idx].cpu().numpy()
    a_sm = tbl.act[idx].cpu().numpy()
    want_R = d[order, 3].astype(np.float32 if storage == torch.float32 else np.float64)
    assert np.array_equal(R_sm, want_R)
    assert np.array_equal(a_sm, ac[order].astype(np.uint8))
    # padding is zero, so the data elements are all there is
    total = int(rows.sum()) * 64
    if total:
        Rn = tbl.R[:total].cpu().numpy()
        assert np.count_nonzero(Rn) == np.count_nonzero(want_R)
        assert int(tbl.act[:total].to(torch.int64).sum()) == int(ac.sum())
    if arrival:
        assert np.array_equal(tbl.rec_state.cpu().numpy(), st)
        t_ref = np.empty(N, dtype=np.int64)
        off = np.concatenate([[0], np.cumsum(counts)])
        t_ref[order] = np.arange(N) - off[st[order]]
        assert np.array_equal(tbl.rec_t.cpu().numpy(), t_ref)
        e_ref = tbl.elem(torch.from_numpy(st).to(tbl.device), torch.from_numpy(t_ref).to(tbl.device)).cpu().numpy()
        assert np.array_equal(tbl.rec_elem.cpu().numpy(), e_ref)
        assert np.array_equal(tbl.R[tbl.rec_elem].cpu().numpy(), d[:, 3].astype(want_R.dtype))
    else:
        assert tbl.rec_elem is None and tbl.rec_t is None and tbl.rec_state is None
    return tbl


@pytest.mark.parametrize("kind", ["uniform", "skewed", "one_state", "state_major", "reversed", "round_robin"])
@pytest.mark.parametrize("S,A,N", [(1, 30, 20000), (20, 11, 20000), (64, 11, 5000), (65, 3, 9000), (300, 32, 40000),
                                   (5000, 11, 70001), (70000, 16, 300000)])
def test_ingest_vs_stable_numpy_sort(dc, kind, S, A, N):
    rng = np.random.default_rng(stable_seed(kind, S, N))
    d = make_table(rng, N, S, A, kind)
    check_table(dc, d, S, A, torch.float32)


@pytest.mark.parametrize("pairs", ["1", "0"])
@pytest.mark.parametrize("kind", ["uniform", "skewed", "one_state", "state_major", "reversed", "round_robin"])
@pytest.mark.parametrize("S,A,N", [(1, 30, 20000), (20, 11, 20000), (65, 3, 9000), (300, 32, 40000), (5000, 11, 70001),
                                   (70000, 16, 300000), (256, 11, 1_000_003)])
def test_ingest_pair_records_vs_stable_numpy_sort(dc, kind, S, A, N, pairs, knob):
    _pair_records_case(dc, kind, S, A, N, pairs, knob, arrival=False)


@pytest.mark.parametrize("pairs", ["1", "0"])
@pytest.mark.parametrize("kind", ["uniform", "skewed", "state_major", "round_robin"])
@pytest.mark.parametrize("S,A,N", [(1, 30, 20000), (20, 11, 20000), (300, 32, 40000), (5000, 11, 70001), (70000, 16, 300000),
                                   (256, 11, 1_000_003)])
def test_ingest_pair_records_with_arrival_bookkeeping(dc, kind, S, A, N, pairs, knob):
    """The same with rec_elem / rec_t / rec_state: the pair passes log where every record goes (one coalesced word per record
    and pass) and the arrival -> element map is the logs composed — one, two, three (70 000 states) and four (2e7 states) passes."""
    _pair_records_case(dc, kind, S, A, N, pairs, knob, arrival=True)


def test_ingest_four_pass_arrival_logs(dc):
    """2e7 states = 25 key bits = four passes: all four position logs of the arrival bookkeeping are composed."""
    rng = np.random.default_rng(99)
    d = make_table(rng, 250_000, 20_000_000, 11, "uniform")
    check_table(dc, d, 20_000_000, 11, torch.float32, arrival=True)


def _pair_records_case(dc, kind, S, A, N, pairs, knob, arrival):
    """f32 tables without arrival bookkeeping travel as 8-byte {key, value} records through the whole-line passes
    (rx_scatter_lines_kernel: stores in 64-byte units, what is left of a digit waits in registers for the next tile);
    DCARL_INGEST_PAIRS=0 keeps them on the two-array passes.  Both against the stable NumPy sort."""
    if pairs != "1":
        knob("DCARL_INGEST_PAIRS", pairs)
    rng = np.random.default_rng(stable_seed(kind, S, N, 5))
    d = make_table(rng, N, S, A, kind)
    check_table(dc, d, S, A, torch.float32, arrival=arrival)
    check_table(dc, d, S, A, torch.float32, sort_by_length=False, arrival=arrival)


# ---- the direct path (round 4): tile-partitioned compact records -> count -> pack straight into the sliced layout -------------
@pytest.mark.parametrize("kind", ["uniform", "skewed", "one_state", "state_major", "reversed", "round_robin"])
@pytest.mark.parametrize("S,A,N", [(1, 30, 20000), (20, 11, 20000), (64, 11, 5000), (65, 3, 9000), (255, 5, 30000), (256, 11, 1_000_003),
                                   (257, 2, 70000), (300, 32, 40000), (5000, 11, 70001), (65536, 16, 300000), (40000, 11, 2_000_000)])
def test_ingest_direct_path_vs_stable_numpy_sort(dc, kind, S, A, N, monkeypatch):
    """DCARL_INGEST_DIRECT=1 forces the direct path at every size (the default takes it from 2^20 records): the same table, bit for
    bit, as the stable NumPy sort — lengths, slot order, row offsets, every state's records in arrival order, zero padding — for
    arrival orders from uniform to state-major (where ONE bucket receives whole tiles and a group's stream runs to many
    chunks), with and without sorted slots."""
    monkeypatch.setenv("DCARL_INGEST_DIRECT", "1")
    rng = np.random.default_rng(stable_seed(kind, S, N, 9))
    d = make_table(rng, N, S, A, kind)
    check_table(dc, d, S, A, torch.float32, arrival=False)
    check_table(dc, d, S, A, torch.float32, sort_by_length=False, arrival=False)
    # the sort path on the same table gives the identical buffers
    a = dc.RecordTable.from_reference_table(d, S, A, arrival=False)
    monkeypatch.setenv("DCARL_INGEST_DIRECT", "0")
    b = dc.RecordTable.from_reference_table(d, S, A, arrival=False)
    assert torch.equal(a.R, b.R) and torch.equal(a.act, b.act) and torch.equal(a.lengths, b.lengths) and torch.equal(a.slice_row_off, b.slice_row_off)


@pytest.mark.parametrize("kind", ["skewed", "state_major", "uniform"])
def test_ingest_direct_path_more_items_than_persistent_blocks(dc, kind, monkeypatch):
    """9.4e6 records of 65 536 states = 12 groups x 256 buckets = 3 072 (bucket, group) items for the 1 024 (pack) / 2 048 (count)
    persistent blocks: every block loops over several items of its XCD's queue and, on the skewed table, takes other XCDs' items."""
    monkeypatch.setenv("DCARL_INGEST_DIRECT", "1")
    rng = np.random.default_rng(31)
    d = make_table(rng, 6656 * 128 * 11 + 77, 65536, 11, kind)
    check_table(dc, d, 65536, 11, torch.float32, arrival=False)


@pytest.mark.parametrize("count", ["queue", "wide"])
@pytest.mark.parametrize("kind", ["uniform", "skewed", "state_major", "one_state", "round_robin"])
@pytest.mark.parametrize("S,N", [(300, 50_000), (5000, 70_001), (8192, 900_000), (16384 + 77, 1_200_003), (65536, 6656 * 130 + 5)])
def test_ingest_direct_path_both_count_passes(dc, count, kind, S, N, monkeypatch, knob):
    """The direct path's count pass has two forms — an item per (bucket, group) on persistent blocks (tables of fewer than 32 buckets) and a
    block per (group, 64 buckets) with lane = bucket and a transposed counter table (everything larger; runs of more than 64 bytes, as
    on skewed and state-major tables, go through the wave's own table) — DCARL_DP_COUNT forces either at every size: the same table
    bit for bit, including bucket ranges that are partly filled (16 461 states = 65 buckets: the second range holds one)."""
    monkeypatch.setenv("DCARL_INGEST_DIRECT", "1")
    knob("DCARL_DP_COUNT", count)
    rng = np.random.default_rng(stable_seed(kind, S, N, 77))
    d = make_table(rng, N, S, 11, kind)
    check_table(dc, d, S, 11, torch.float32, arrival=False)


@pytest.mark.parametrize("N", [0, 1, 3, 4, 5, 63, 64, 65, 6655, 6656, 6657, 13312, 13313, 6656 * 128 - 1, 6656 * 128, 6656 * 128 + 1, 6656 * 256 + 1, 6656 * 300 + 17])
def test_ingest_direct_path_tile_and_group_edges(dc, N, monkeypatch):
    """Tile (6 656 records) and group (128 tiles) edges of the direct path."""
    monkeypatch.setenv("DCARL_INGEST_DIRECT", "1")
    rng = np.random.default_rng(N + 23)
    for S in (1, 7, 200, 5000):
        d = make_table(rng, N, S, 11, "uniform")
        check_table(dc, d, S, 11, torch.float32, arrival=False)


def test_ingest_direct_path_is_the_default_for_large_f32_tables(dc, monkeypatch):
    """Without the variable: 2^20 records and more of an f32 table of 2 048 .. 65 536 states without arrival bookkeeping take the direct path (seen in the
    workspace size: ONE record buffer instead of two), everything else the sort; results as above."""
    monkeypatch.delenv("DCARL_INGEST_DIRECT", raising=False)
    lib = dc.load_library()
    N, S, A = (1 << 20) + 5, 3000, 11
    assert lib.dcarl_ingest_workspace_bytes(N, S, A, 4, 0, 0) == lib.dcarl_ingest_workspace_bytes(N, S, A, 4, 8, 0)   # automatic == forced here
    assert lib.dcarl_ingest_workspace_bytes(N, S, A, 4, 0, 0) < 0.7 * lib.dcarl_ingest_workspace_bytes(N, S, A, 4, 4, 0)   # one record buffer, not two
    assert lib.dcarl_ingest_workspace_bytes(N - 10, S, A, 4, 0, 0) == lib.dcarl_ingest_workspace_bytes(N - 10, S, A, 4, 4, 0)   # below 2^20 records: sort
    assert lib.dcarl_ingest_workspace_bytes(N, 2000, A, 4, 0, 0) == lib.dcarl_ingest_workspace_bytes(N, 2000, A, 4, 4, 0)   # below 2 048 states: sort
    assert lib.dcarl_ingest_workspace_bytes(N, 70000, A, 4, 8, 0) == lib.dcarl_ingest_workspace_bytes(N, 70000, A, 4, 4, 0)   # > 65 536 states: sort
    rng = np.random.default_rng(4)
    d = make_table(rng, N, S, A, "skewed")
    check_table(dc, d, S, A, torch.float32, arrival=False)
    check_table(dc, d, S, A, torch.float32, arrival=True)            # arrival bookkeeping: the sort path
    check_table(dc, d, S, A, torch.float64, arrival=False)           # f64 storage: the sort path


@pytest.mark.parametrize("N", [0, 1, 3, 7, 8, 9, 63, 64, 65, 6655, 6656, 6657, 13312, 16383, 16384, 16385, 40000])
def test_ingest_pair_records_small_and_tile_edges(dc, N):
    rng = np.random.default_rng(N + 11)
    for S in (1, 7, 200, 5000):
        d = make_table(rng, N, S, 11, "uniform")
        check_table(dc, d, S, 11, torch.float32, arrival=False)
        check_table(dc, d, S, 11, torch.float32, arrival=True)


@pytest.mark.parametrize("storage", [torch.float32, torch.float64])
@pytest.mark.parametrize("sort_by_length", [True, False])
@pytest.mark.parametrize("arrival", [True, False])
def test_ingest_options(dc, storage, sort_by_length, arrival):
    rng = np.random.default_rng(7)
    d = make_table(rng, 123457, 1000, 11, "skewed")
    check_table(dc, d, 1000, 11, storage, sort_by_length, arrival)


@pytest.mark.parametrize("N", [0, 1, 3, 63, 64, 65, 8191, 8192, 8193, 16385])
def test_ingest_small_and_tile_edges(dc, N):
    rng = np.random.default_rng(N)
    for S in (1, 7, 200):
        d = make_table(rng, N, S, 11, "uniform")
        check_table(dc, d, S, 11, torch.float64)


@pytest.mark.parametrize("threads", ["256", "512"])
def test_ingest_both_scatter_instances(dc, threads, knob):
    """The ranked scatter has a 256-thread (tile of 4 096) and a 512-thread (tile of 8 192) instance; the launcher picks by
    block length (the short one only from ~2.7e8 records on), so both are forced here."""
    knob("DCARL_INGEST_SCATTER_THREADS", threads)
    rng = np.random.default_rng(int(threads))
    for S, N, kind in ((70000, 300001, "uniform"), (300, 50000, "state_major"), (5000, 123456, "skewed")):
        d = make_table(rng, N, S, 11, kind)
        check_table(dc, d, S, 11, torch.float32)
        check_table(dc, d, S, 11, torch.float64, arrival=False)


def test_ingest_many_blocks(dc):
    """More than one block per pass and more than one tile per block (N > 2048 tiles of 8192 needs 16.8e6 records)."""
    rng = np.random.default_rng(3)
    N, S = 17_000_000, 40_000
    d = make_table(rng, N, S, 11, "uniform")
    check_table(dc, d, S, 11, torch.float32, arrival=False)


def test_ingest_device_table_and_limit(dc, sim2_data):
    data = sim2_data[0]
    t = torch.from_numpy(data).cuda()
    tbl = dc.RecordTable.from_reference_table(t, 20, 11, storage=torch.float64, limit=20000)
    assert tbl.n_records == 20000
    ref = check_table(dc, data[:20000], 20, 11, torch.float64)
    assert torch.equal(tbl.R, ref.R) and torch.equal(tbl.act, ref.act) and torch.equal(tbl.rec_elem, ref.rec_elem)
    assert torch.equal(tbl.state_feature, torch.from_numpy(data[:20000, 1]).cuda())


@pytest.mark.parametrize("S,A,N,kind", [(1, 1, 100, "uniform"), (1, 11, 20000, "uniform"), (20, 11, 49866, "skewed"),
                                        (100, 32, 100000, "uniform"), (3000, 16, 250000, "skewed"), (3000, 7, 0, "uniform"),
                                        (70000, 11, 400000, "state_major"), (65536, 11, 3_000_000, "uniform"), (1, 1, 5000, "uniform")])
@pytest.mark.parametrize("storage,pairs", [(torch.float32, "1"), (torch.float32, "0"), (torch.float64, "1")])
def test_ingest_buckets_vs_numpy(dc, S, A, N, kind, storage, pairs, knob):
    """The final-state layout straight from the arrival-ordered table == the reference's data_state_act (S1:80).  (f32: through
    the pair-record passes, whose last one writes the values alone and reports the bucket bounds, or the two-array passes.)"""
    if pairs != "1":
        knob("DCARL_INGEST_PAIRS", pairs)
    import ctypes as C
    from dcarl_amd import _lib
    from dcarl_amd.records import as_device_table, check_ingest_info
    lib = dc.load_library()
    dev = dc.require_gpu()
    rng = np.random.default_rng(N + S)
    d = make_table(rng, N, S, A, kind)
    dd = as_device_table(d, dev)
    f32 = storage == torch.float32
    ws = torch.empty(int(lib.dcarl_ingest_workspace_bytes(N, S, A, 4 if f32 else 8, 0, 1)), dtype=torch.uint8, device=dev)
    vals = torch.zeros(max(N, 4), dtype=storage, device=dev)
    seg = torch.empty(S * A + 1, dtype=torch.int64, device=dev)
    info = torch.empty(16, dtype=torch.int64, device=dev)
    fn = lib.dcarl_ingest_buckets_f32 if f32 else lib.dcarl_ingest_buckets_f64
    _lib.check(fn(_lib.ptr(dd), N, S, A, _lib.ptr(ws), _lib.ptr(vals), _lib.ptr(seg), _lib.ptr(info), _lib.stream_ptr()))
    check_ingest_info(info, S, A, N)
    key = d[:, 0].astype(np.int64) * A + d[:, 2].astype(np.int64)
    order = np.argsort(key, kind="stable")
    want_seg = np.concatenate([[0], np.cumsum(np.bincount(key, minlength=S * A))])
    assert np.array_equal(seg.cpu().numpy(), want_seg)
    assert np.array_equal(vals[:N].cpu().numpy(), d[order, 3].astype(np.float32 if f32 else np.float64))


def test_bounds_from_reference_table_matches_the_online_table(dc, sim2_data):
    est = dc.ConfidenceEstimator()
    data = sim2_data[0][:20000]
    b = est.bounds_from_reference_table(data, 20, 11, storage=torch.float64)
    tr = est.trace(dc.RecordTable.from_reference_table(data, 20, 11, storage=torch.float64))
    assert torch.equal(b.n, tr.n) and torch.equal(b.amax, tr.amax)
    assert (b.V - tr.V).abs().max().item() < 1e-9


def test_ingest_refuses_bad_tables(dc):
    good = np.array([[0, 0.5, 1, 3.0], [4, 0.5, 10, 3.0]])
    dc.RecordTable.from_reference_table(good, 5, 11)
    for bad, exc in ((np.array([[0, 0.5, 1, 3.0], [7, 0.5, 1, 3.0]]), IndexError),
                     (np.array([[-1, 0.5, 1, 3.0]]), IndexError),
                     (np.array([[0, 0.5, 11, 3.0]]), IndexError),
                     (np.array([[0, 0.5, -2, 3.0]]), IndexError),
                     (np.array([[np.nan, 0.5, 1, 3.0]]), ValueError),
                     (np.array([[0, 0.5, np.inf, 3.0]]), ValueError),
                     (np.array([[0, 0.5, 1, np.nan]]), ValueError),
                     (np.array([[0, 0.5, 1, -np.inf]]), ValueError)):
        with pytest.raises(exc):
            dc.RecordTable.from_reference_table(bad, 5, 11)
        with pytest.raises(exc):
            dc.ConfidenceEstimator().bounds_from_reference_table(bad, 5, 11)
    # finite in float64 but not in float32: refused for f32 storage only
    big = np.array([[0, 0.5, 1, 1e300]])
    with pytest.raises(ValueError):
        dc.RecordTable.from_reference_table(big, 5, 11, storage=torch.float32)
    dc.RecordTable.from_reference_table(big, 5, 11, storage=torch.float64)
    with pytest.raises(ValueError):
        dc.RecordTable.from_reference_table(np.zeros((3, 3)), 1, 11)
    # -0.5 truncates to state 0 like int() does (S1:77)
    t = dc.RecordTable.from_reference_table(np.array([[-0.5, 0.5, 1.9, 3.0]]), 5, 11)
    assert int(t.lengths_by_state[0]) == 1 and int(t.act[0]) == 1


@pytest.mark.parametrize("storage", [torch.float32, torch.float64])
def test_export_then_ingest_round_trip(dc, storage):
    """to_reference_table is the inverse of from_reference_table: rows in arrival order -> table -> the same rows."""
    rng = np.random.default_rng(11)
    d = make_table(rng, 60001, 333, 11, "skewed")
    d[:, 0] = np.floor(d[:, 0])
    states = rng.random(333)
    d[:, 1] = states[d[:, 0].astype(np.int64)]
    d[:, 3] = d[:, 3].astype(np.float32 if storage == torch.float32 else np.float64)
    tbl = dc.RecordTable.from_reference_table(d, 333, 11, storage=storage)
    back = tbl.to_reference_table(states=states)
    assert np.array_equal(back.cpu().numpy(), d)
    with pytest.raises(ValueError):
        tbl.to_reference_table(dense_order=True)                    # ragged
    with pytest.raises(ValueError):
        dc.RecordTable.from_reference_table(d, 333, 11, arrival=False).to_reference_table()


@pytest.mark.parametrize("S,T", [(1, 20000), (64, 100), (192, 1001), (1000, 64), (4096, 50)])
def test_dense_order_export_is_a_permutation_that_regroups_exactly(dc, S, T):
    """The synthetic arrival order bench.py's end-to-end workload uses: every record appears once, a state's records keep
    their order, and ingesting the exported rows gives the identical table back (encode -> decode round trip)."""
    q = torch.linspace(-50, 100, 11)
    tbl = dc.sampler.sample_state_records(q, T, seed=3, S=S)
    d = tbl.to_reference_table(dense_order=True)
    dn = d.cpu().numpy()
    assert dn.shape == (S * T, 4)
    st = dn[:, 0].astype(np.int64)
    assert np.array_equal(np.bincount(st, minlength=S), np.full(S, T))
    for t in (0, T // 2, T - 1):                                    # every state exactly once per round
        assert np.array_equal(np.sort(st[t * S:(t + 1) * S]), np.arange(S))
    if S > 1 and T > 1:
        assert not np.array_equal(st[:S], st[S:2 * S])              # and the order inside a round changes
    t2 = dc.RecordTable.from_reference_table(d, S, 11, arrival=False)
    assert torch.equal(t2.R, tbl.R) and torch.equal(t2.act, tbl.act) and torch.equal(t2.lengths, tbl.lengths)


@pytest.mark.parametrize("S", [1, 64, 65, 1000, 70001])
def test_slot_order_vs_numpy(dc, S):
    """dcarl_slot_order (what from_state_major and the ragged sampler use instead of a torch sort): states by descending length,
    stable, and the slice row offsets of that order."""
    from dcarl_amd.records import slot_order
    rng = np.random.default_rng(S)
    for hi in (1, 5, 3000, 2_000_000):
        lens = rng.integers(0, hi, S)
        if S > 10:
            lens[rng.integers(0, S, S // 4)] = lens[0]                 # plenty of ties: stability matters
        len_slot, slot_state, state_slot, sro, rows = slot_order(torch.from_numpy(lens))
        if S > 64:
            want = np.argsort(-lens, kind="stable")
            assert np.array_equal(slot_state.cpu().numpy(), want)
            inv = np.empty(S, dtype=np.int64); inv[want] = np.arange(S)
            assert np.array_equal(state_slot.cpu().numpy(), inv)
            sl = lens[want]
        else:
            assert slot_state is None and state_slot is None
            sl = lens
        assert np.array_equal(len_slot.cpu().numpy(), sl)
        W = (S + 63) // 64
        pad = np.zeros(W * 64, dtype=np.int64); pad[:S] = sl
        r = (pad.reshape(W, 64).max(1) + 3) // 4 * 4
        assert np.array_equal(sro.cpu().numpy(), np.concatenate([[0], np.cumsum(r)])) and rows == int(r.sum())
        ls2, ss2, _, sro2, _ = slot_order(torch.from_numpy(lens), sort_by_length=False)
        assert ss2 is None and np.array_equal(ls2.cpu().numpy(), lens)


def test_ingest_pack_refuses_a_workspace_grouped_with_other_arguments(dc):
    """The pack call rebuilds the workspace layout from N / S / A / flags / the element width: what differs from the group call
    of the same workspace is refused (it would misread the workspace), the matching call packs the table."""
    from dcarl_amd import _lib, layout, records
    lib = dc.load_library()
    rng = np.random.default_rng(77)
    N, S, A = 5000, 300, 11
    d = torch.as_tensor(make_table(rng, N, S, A, "uniform")).cuda()
    flags = records.INGEST_SORT_BY_LENGTH | records.INGEST_NO_DIRECT
    ws = torch.empty(int(lib.dcarl_ingest_workspace_bytes(N, S, A, 4, flags | records.INGEST_FORCE_DIRECT, 0)) * 2, dtype=torch.uint8, device="cuda")
    W = layout.num_slices(S)
    lengths, slot_state, state_slot = (torch.empty(S, dtype=torch.int32, device="cuda") for _ in range(3))
    sro = torch.empty(W + 1, dtype=torch.int64, device="cuda")
    info = torch.empty(records.INGEST_INFO_WORDS, dtype=torch.int64, device="cuda")
    assert lib.dcarl_ingest_group_f32(_lib.ptr(d), N, S, A, flags, _lib.ptr(ws), _lib.ptr(lengths), _lib.ptr(slot_state), _lib.ptr(state_slot),
                                      _lib.ptr(sro), None, _lib.ptr(info), _lib.stream_ptr()) == 0
    rows, bands, _ = records.check_ingest_info(info, S, A, N)
    R = torch.empty(rows * 64, dtype=torch.float32, device="cuda")
    act = torch.empty(rows * 64, dtype=torch.uint8, device="cuda")

    def pack(fn, n, s, a, fl, r):
        return fn(n, s, a, fl, _lib.ptr(ws), _lib.ptr(lengths), _lib.ptr(slot_state), _lib.ptr(sro), bands, _lib.ptr(r), _lib.ptr(act), None, None,
                  _lib.stream_ptr())
    f32 = lib.dcarl_ingest_pack_f32
    for args in ((N - 1, S, A, flags), (N, S + 1, A, flags), (N, S, A + 1, flags), (N, S, A, flags & ~records.INGEST_SORT_BY_LENGTH),
                 (N, S, A, (flags & ~records.INGEST_NO_DIRECT) | records.INGEST_FORCE_DIRECT)):
        assert pack(f32, *args, R) == -1 and b"was grouped with" in lib.dcarl_last_error(), args
    assert pack(lib.dcarl_ingest_pack_f64, N, S, A, flags, torch.empty(rows * 64, dtype=torch.float64, device="cuda")) == -1
    assert pack(f32, N, S, A, flags, R) == 0
    torch.cuda.synchronize()
    ref = dc.RecordTable.from_reference_table(d, S, A, arrival=False)
    assert torch.equal(R, ref.R[:rows * 64]) and torch.equal(act, ref.act[:rows * 64])


# ---- round 5: data_state_act (the (state, action) bucket layout) without the two-pass sort ---------------------------------------
@pytest.mark.parametrize("kind", ["uniform", "skewed", "state_major", "round_robin"])
@pytest.mark.parametrize("S,A,N", [(1, 30, 20000), (20, 11, 49866), (300, 32, 40000), (2048, 11, 1_200_003), (5000, 16, 2_000_000),
                                   (65536, 11, 3_000_000), (40000, 5, 1_048_576)])
def test_buckets_regroup_route_equals_the_sort_route_bit_for_bit(dc, kind, S, A, N):
    """records.buckets_from_reference_table: the table into the sliced layout (direct ingest where it qualifies) + dcarl_group_records_*
    (every state's stream regrouped by action in LDS-staged chunks) gives the arrays of dcarl_ingest_buckets_* (the stable two-pass
    radix sort by (state, action)) bit for bit — values, offsets — and both equal the stable NumPy sort, i.e. the reference's
    data_state_act[idx][act] lists (S1:80) in order."""
    from dcarl_amd.records import buckets_from_reference_table
    rng = np.random.default_rng(stable_seed(kind, S, N, 505))
    d = make_table(rng, N, S, A, kind)
    v1, s1 = buckets_from_reference_table(d, S, A, via="regroup")
    v2, s2 = buckets_from_reference_table(d, S, A, via="sort")
    assert torch.equal(s1, s2) and int(s1[-1]) == N
    assert torch.equal(v1[:N], v2[:N])
    key = d[:, 0].astype(np.int64) * A + d[:, 2].astype(np.int64)
    order = np.argsort(key, kind="stable")
    assert np.array_equal(v1[:N].cpu().numpy(), d[order, 3].astype(np.float32))
    assert np.array_equal(s1.cpu().numpy(), np.concatenate([[0], np.cumsum(np.bincount(key, minlength=S * A))]))
    va, sa = buckets_from_reference_table(d, S, A)                 # auto: regroup for tables the direct ingest takes, the sort otherwise
    assert torch.equal(sa, s2) and torch.equal(va[:N], v2[:N])


@pytest.mark.parametrize("storage", [torch.float32, torch.float64])
def test_buckets_regroup_route_f64_and_limit(dc, sim2_data, storage):
    from dcarl_amd.records import buckets_from_reference_table
    data = sim2_data[0]
    v1, s1 = buckets_from_reference_table(data, 20, 11, storage=storage, limit=20000, via="regroup")
    v2, s2 = buckets_from_reference_table(data, 20, 11, storage=storage, limit=20000, via="sort")
    assert torch.equal(s1, s2) and torch.equal(v1[:20000], v2[:20000]) and v1.dtype == storage
    est = dc.ConfidenceEstimator()
    a = est.bounds_from_reference_table(data, 20, 11, storage=storage, limit=20000, via="buckets")
    b = est.bounds_from_reference_table(data, 20, 11, storage=storage, limit=20000, via="online")
    assert torch.equal(a.amax, b.amax) and torch.equal(a.n, b.n) and float((a.V - b.V).abs().max()) <= 1e-9
    with pytest.raises(ValueError):
        buckets_from_reference_table(data, 20, 11, via="nope")
