"""Pin the C oracle (oracle/dcarl_oracle.c) against the reference goldens and the NumPy oracle."""
import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import dcarl_oracle as orc


def group_by_state(data, S, limit=20000):
    d = data[:limit]
    st = d[:, 0].astype(np.int64)
    order = np.argsort(st, kind="stable")
    off = np.zeros(S + 1, np.int64)
    off[1:] = np.cumsum(np.bincount(st, minlength=S))
    return d[order, 3].copy(), d[order, 2].astype(np.uint8), off, st, order


@pytest.mark.parametrize("name,S,A", [("sim1_trace.npz", 1, 30), ("sim2_trace.npz", 20, 11)])
@pytest.mark.parametrize("recompute", [False, True])
def test_c_trace_matches_reference(golden, sim1_data, sim2_data, name, S, A, recompute):
    data = (sim1_data if S == 1 else sim2_data)[0]
    g = golden(name)
    R, act, off, st, order = group_by_state(data, S)
    res = co.trace(R, act, off, S, A, recompute=recompute)
    assert np.array_equal(res["step_act"], g["step_act"])
    assert np.max(np.abs(res["step_val"] - g["step_value"])) <= 1e-11
    assert np.array_equal(res["activation_step"], g["activation_step"])
    if not recompute:
        assert np.max(np.abs(res["V"] - g["TSRL_value"])) <= 1e-11
        assert np.array_equal(res["n"], g["bucket_len"])
    if S == 20 and not recompute:
        pos = np.empty(len(order), np.int64)
        pos[order] = np.arange(len(order))
        ov = co.overall(res["step_val"], res["activation_step"], off, st.astype(np.int32), pos)
        assert np.max(np.abs(ov - g["overall_value"])) <= 1e-9
        assert abs(ov[-1] - 597.7193818873668) <= 1e-9


def test_c_bounds_csr_matches_reference(golden, sim2_data):
    data = sim2_data[0][:20000]
    g = golden("sim2_trace.npz")
    key = data[:, 0].astype(int) * 11 + data[:, 2].astype(int)
    order = np.argsort(key, kind="stable")
    seg = np.zeros(221, np.int64)
    seg[1:] = np.cumsum(np.bincount(key, minlength=220))
    res = co.bounds_csr(data[order, 3].copy(), seg, 20, 11)
    assert np.max(np.abs(res["V"] - g["TSRL_value"])) <= 1e-11
    assert np.array_equal(res["n"], g["bucket_len"])
    assert np.array_equal(res["amax"], np.argmax(g["TSRL_value"], axis=1))


def test_c_bounds_random_buckets(golden):
    g = golden("bounds_random.npz")
    x, off = g["x"], g["off"]
    nb = len(off) - 1
    # two states-worth trick: treat every bucket as action 1 of its own 2-action state (action 0 empty)
    seg = np.zeros(2 * nb + 1, np.int64)
    seg[1::2] = off[:-1]
    seg[2::2] = off[1:]
    res = co.bounds_csr(x, seg, nb, 2)
    ref = np.minimum(g["lower"], g["ci_lower"])
    assert np.max(np.abs(res["V"][:, 1] - ref) / np.maximum(1, np.abs(ref))) <= 1e-11
    # and as the rule action (action 0) of a 1-action state
    res0 = co.bounds_csr(x, off, nb, 1)
    assert np.max(np.abs(res0["V"][:, 0] - g["upper"]) / np.maximum(1, np.abs(g["upper"]))) <= 1e-11


def test_c_philox_and_samplers():
    assert co.philox((0, 0, 0, 0), (0, 0)) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    assert co.philox((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == \
        (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)
    q = np.random.RandomState(1).uniform(-50, 100, (20, 11))
    a1, r1 = co.sample_state_records(q, 777, seed=0x1234567890)
    a2, r2, _ = orc.sample_state_records(q, 777, seed=0x1234567890)
    assert np.array_equal(a1, a2) and np.max(np.abs(r1 - r2)) < 1e-9
    i1, c1, p1 = co.sample_pairs(q, 5000, seed=5, offset=(1 << 32) - 100)
    i2, c2, p2, ok = orc.sample_pairs(q, 5000, seed=5, offset=(1 << 32) - 100)
    assert np.array_equal(i1 >= 0, ok) and np.array_equal(i1[ok], i2[ok]) and np.array_equal(c1, c2)
    assert np.max(np.abs(p1 - p2)) < 1e-9
    for off in (0, 1, 2, 3, 5):                                   # group boundaries / unaligned offsets
        i1, c1, p1 = co.sample_pairs(q, 23, seed=9, offset=off)
        i2, c2, p2, ok = orc.sample_pairs(q, 23, seed=9, offset=off)
        assert np.array_equal(c1, c2) and np.array_equal(i1 >= 0, ok) and np.max(np.abs(p1 - p2)) < 1e-9
        j1, d1, r1 = co.sample_pairs(q, 10, seed=9, offset=off + 7)     # a window of the same stream
        assert np.array_equal(d1, c1[7:17]) and np.array_equal(r1, p1[7:17])


def test_c_trace_f32_inputs_ragged_and_empty():
    rng = np.random.RandomState(0)
    S, A = 37, 16
    lens = rng.randint(0, 400, S)
    lens[3] = 0
    off = np.zeros(S + 1, np.int64)
    off[1:] = np.cumsum(lens)
    N = int(off[-1])
    act = rng.randint(0, A, N).astype(np.uint8)
    R = (rng.uniform(-50, 100, (S, A))[np.repeat(np.arange(S), lens), act] + 50 * rng.standard_normal(N)).astype(np.float32)
    res = co.trace(R, act, off, S, A)
    st = np.repeat(np.arange(S), lens)
    ref = orc.run_online_sums(st, act, R.astype(np.float64), S, A)
    assert np.array_equal(res["step_act"], ref["step_act"])
    assert np.max(np.abs(res["step_val"] - ref["step_val"])) <= 1e-10
    assert np.array_equal(res["activation_step"], ref["activation_step"])
    assert res["activation_step"][3] == -1 and res["amax"][3] == 0 and res["vmax"][3] == 100
