"""Host-side logic that needs no GPU: the figure helper of the Sim2 drop-in, bench.py's self-launch, the shared id check,
and the oracle's restatements of the round-2 samplers against its own Philox."""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import c_oracle as co          # noqa: E402


def test_plot_states_figures():
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    from dcarl_amd.reference_api import plot_states
    plt.close("all")
    rng = np.random.RandomState(0)
    lens = rng.randint(5, 60, 12)
    latch = np.where(rng.rand(12) < 0.5, -1, rng.randint(1, 5, 12))
    g = dict(step_TSRL_value=[rng.rand(n).tolist() for n in lens], activation_step=latch,
             sorted_state_data_len=np.stack([np.arange(12), lens], 1)[np.argsort(-lens)])
    figs = plot_states(g, plt=plt)
    assert [f.number for f in figs] == [1, 2, 3] and plt.get_fignums() == [1, 2, 3]          # 12 states, 5 panels each
    assert [len(f.axes) for f in figs] == [5, 5, 2]
    order = g["sorted_state_data_len"]
    for rank, (sid, n) in enumerate(order.tolist()):
        ax = figs[rank // 5].axes[rank % 5]
        assert ax.get_xlim() == (0.0, float(order[0][1]))
        lines = ax.get_lines()
        if latch[sid] == -1:
            assert len(lines) == 1 and lines[0].get_color() == "darkgray" and len(lines[0].get_ydata()) == n
        else:
            assert [l.get_color() for l in lines] == ["darkgray", "black"]
            assert len(lines[0].get_ydata()) == latch[sid] and list(lines[1].get_xdata()) == list(range(latch[sid], n))
    plt.close("all")


def test_bench_self_launch_command(monkeypatch):
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as ex:
        bench.init_dist(4)
    assert ex.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # under torchrun (WORLD_SIZE set) it must NOT re-launch
    monkeypatch.setenv("WORLD_SIZE", "4")
    seen.clear()
    monkeypatch.setattr(bench.torch.cuda, "set_device", lambda d: (_ for _ in ()).throw(RuntimeError("stop here")))
    with pytest.raises(RuntimeError, match="stop here"):
        bench.init_dist(4)
    assert not seen
    # workload aliases of round 1 still parse
    monkeypatch.setattr(sys, "argv", ["bench.py", "--workload", "sim2_ragged_batch"])
    assert bench.parse().workload == "cfg3_sim2_argmax"
    # algorithmic bytes: SURVEY 8(d) formulas
    assert bench.batch_algorithmic_bytes(1000, 10, 11, True) == 4 * 1000 + 10 * (12 * 11 + 8) + 8 * 111
    assert bench.batch_algorithmic_bytes(1000, 10, 16, False) == 4 * 1000 + 10 * (12 * 16 + 8)


def test_check_ids():
    from dcarl_amd.records import check_ids
    check_ids(torch.tensor([0, 3]), torch.tensor([0, 10]), 4, 11)
    check_ids(None, torch.zeros(0, dtype=torch.int64), 4, 11)
    for st, ac in (([4], [0]), ([-1], [0]), ([0], [11]), ([0], [-2]), ([0], [300])):
        with pytest.raises(IndexError):
            check_ids(torch.tensor(st), torch.tensor(ac), 4, 11)


def test_oracle_ragged_sampler_is_the_dense_one_cut_short():
    q = np.random.RandomState(0).uniform(-50, 100, (6, 11))
    lens = [0, 5, 17, 1, 40, 3]
    a, R, off = co.sample_state_records_ragged(q, lens, seed=77, stream=3)
    a_d, R_d = co.sample_state_records(q, 40, seed=77, stream=3)
    for s, n in enumerate(lens):
        assert np.array_equal(a[off[s]:off[s + 1]], a_d[s][:n]) and np.array_equal(R[off[s]:off[s + 1]], R_d[s][:n])
    live = np.array([1, 2, 3, 11, 5, 7], np.int32)
    a2, _, _ = co.sample_state_records_ragged(q, lens, seed=77, stream=3, n_live=live)
    assert (a2 < np.repeat(live, lens)).all() and (a2[off[1]:off[2]] <= 1).all()


def test_oracle_bucket_sampler_counters_and_moments():
    S, A = 3, 2
    seg = np.array([0, 5, 5, 9, 30, 30, 4000])
    q = np.arange(6.0).reshape(S, A) * 10
    v = co.sample_buckets(q, seg, S, seed=5, stream=2, sigma=1.0)
    # sample i uses Philox counter (i/4, 0, stream, 1): words (0,1) for i%4 < 2, (2,3) otherwise; cos for even, sin for odd
    for i in (0, 1, 6, 7, 29, 3999):
        w = co.philox((i // 4, 0, 2, 1), (5, 0))
        u = lambda x: (x + 0.5) / 4294967296.0
        k = i % 4
        rad = np.sqrt(-2 * np.log(u(w[k & 2])))
        th = 2 * np.pi * u(w[(k & 2) + 1])
        z = rad * (np.sin(th) if k & 1 else np.cos(th))
        b = np.searchsorted(seg, i, side="right") - 1
        assert abs(v[i] - (q.ravel()[b] + z)) < 1e-12
    z = v[30:] - 50.0
    assert abs(z.mean()) < 0.06 and abs(z.std() - 1) < 0.05


def test_bucket_to_state_division_by_multiplication():
    """bounds_quad_kernel turns bucket j of a 16-state block into (state, action) with (j * (65536/A + 1)) >> 16."""
    for A in range(1, 33):
        amul = 65536 // A + 1
        for j in range(16 * A + 64):
            assert (j * amul) >> 16 == j // A


def test_narrowing_rule():
    """ConfidenceEstimator._narrowed: only for A > 16, one never-sampled NON-rule stand-in above the sampled range and above
    the rule action (with rule_act == max_action + 1 the last kept candidate must not be the rule action itself)."""
    import dcarl_amd as dc
    from types import SimpleNamespace as T
    est = dc.ConfidenceEstimator()
    assert est._narrowed(T(A=30, max_action=10)) == 12
    assert est._narrowed(T(A=30, max_action=None)) == 30
    assert est._narrowed(T(A=16, max_action=3)) == 16              # the multi-wave kernel serves it anyway
    assert est._narrowed(T(A=30, max_action=-1)) == 2              # no records: the rule action + one stand-in for the rest
    assert est._narrowed(T(A=30, max_action=28)) == 30 and est._narrowed(T(A=30, max_action=29)) == 30
    assert dc.ConfidenceEstimator(dc.Params(rule_act=20))._narrowed(T(A=30, max_action=4)) == 22
    assert dc.ConfidenceEstimator(dc.Params(rule_act=5))._narrowed(T(A=30, max_action=4)) == 7     # 0..4 sampled, rule 5, stand-in 6


def test_graft_entry_has_no_pinned_abi_number():
    """build() compares the library's version with the binding's constant, not with a literal that goes stale."""
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "__graft_entry__.py")).read()
    assert "dcarl_version() == dcarl_amd._lib.ABI_VERSION" in src


def test_ingest_info_read_back_raises_like_the_reference():
    """records.check_ingest_info: what the ingest kernels report (device int64[16]) -> the exceptions the reference would raise
    (IndexError for ids past the table, S1:80) or that this library adds (negative ids, NaN / Inf)."""
    import torch
    from dcarl_amd.records import check_ingest_info

    def info(rows=8, bands=1, maxlen=5, amax=3, smin=0, smax=4, amin=0, flags=0):
        t = torch.zeros(16, dtype=torch.int64)
        t[:8] = torch.tensor([rows, bands, maxlen, amax, smin, smax, amin, flags])
        return t
    assert check_ingest_info(info(), 5, 11, 10) == (8, 1, 3)
    assert check_ingest_info(info(rows=0, bands=0), 5, 11, 0) == (0, 0, -1)          # an empty table reports nothing
    for bad, exc in ((info(smax=5), IndexError), (info(smin=-1), IndexError), (info(amax=11), IndexError), (info(amin=-2), IndexError),
                     (info(flags=1), ValueError), (info(flags=2), ValueError), (info(flags=3), ValueError)):
        with pytest.raises(exc):
            check_ingest_info(bad, 5, 11, 10)
    # a NaN id is reported as INT32_MIN by the kernel: the NaN message wins over the range message
    with pytest.raises(ValueError):
        check_ingest_info(info(smin=-2 ** 31, flags=2), 5, 11, 10)


def test_legacy_stream_switch():
    from dcarl_amd import reference_api as api
    assert api._legacy(None) is True and api._legacy(False) is False
    api.use_legacy_streams(False)
    try:
        assert api._legacy(None) is False and api._legacy(True) is True
    finally:
        api.use_legacy_streams(True)


def test_stream_chunking_of_host_tables():
    """dcarl_amd.stream cuts a host table into consecutive chunks of arrivals (S1:73's front-to-back walk): data[0:limit], ragged
    iterables, empty pieces — host logic, no GPU."""
    from dcarl_amd.stream import _as_chunks
    a = np.arange(40, dtype=np.float64).reshape(10, 4)
    it, total, whole = _as_chunks(a, 4, None)
    parts = list(it)
    assert total == 10 and [p.shape[0] for p in parts] == [4, 4, 2] and np.array_equal(np.concatenate(parts), a) and whole.shape == (10, 4)
    it, total, _ = _as_chunks(a, 4, 6)                            # data[0:6]
    assert total == 6 and [p.shape[0] for p in it] == [4, 2]
    it, total, _ = _as_chunks(torch.from_numpy(a), 100, 1000)     # a CPU tensor; a limit beyond the table is the table
    assert total == 10 and [p.shape[0] for p in it] == [10]
    it, total, whole = _as_chunks(iter([a[:3], a[3:3], torch.from_numpy(a[3:])]), 2, 8)
    parts = list(it)
    assert total is None and whole is None and [p.shape[0] for p in parts] == [2, 1, 2, 2, 1]
    assert np.array_equal(np.concatenate(parts), a[:8])
    for bad in (np.zeros((3, 5)), np.zeros(12), np.zeros((3, 4), dtype=np.float32)):
        with pytest.raises(ValueError):
            _as_chunks(bad, 4, None)
    with pytest.raises(ValueError):
        list(_as_chunks(iter([np.zeros((2, 3))]), 4, None)[0])
