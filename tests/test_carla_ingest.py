"""CARLA record ingest: parser against the reference's example file, grid cells and the
end-to-end path text -> state ids -> record table -> confidence values."""
import os
import sys

import numpy as np
import torch
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
EXAMPLE = os.path.join(REPO, "tests", "golden", "carla_example_collected_data.txt")   # the reference's sample file


def _parse():
    from dcarl_amd import carla_records as cr
    return cr.parse_collected_data(EXAMPLE)


def test_parser_on_the_reference_example_file():
    obs, action, reward = _parse()
    assert obs.shape == (170, 20) and action.shape == (170,) and reward.shape == (170,)
    # the collector cycles used_action over the 10 candidates + brake (dqn_value_collect.py:143-144)
    assert action[:23].tolist() == [i % 11 for i in range(23)]
    assert np.bincount(action).tolist() == [16] * 5 + [15] * 6
    # first record of the file, verbatim
    assert obs[0, 0] == 243.62413025 and obs[0, 19] == -1.54491266 and reward[0] == 18.47598255857919
    # per-action episode returns quoted in SURVEY.md section 8(f)
    m = [reward[action == a].mean() for a in range(11)]
    assert abs(m[0] - 18.39) < 0.01 and abs(m[2] + 79.05) < 0.01 and abs(m[3] - 27.84) < 0.01
    assert abs(reward[action == 0].std() - 0.18) < 0.01


@pytest.mark.gpu
def test_cells_and_end_to_end_confidence_values():
    import torch
    import dcarl_amd as dc
    from dcarl_amd import carla_records as cr
    from oracle import c_oracle as co
    obs, action, reward = _parse()
    cells = cr.state_cells(obs)
    assert np.array_equal(cells.cpu().numpy(), np.floor(obs / np.array(cr.DEFAULT_CELL_WIDTH)).astype(np.int32))
    rng = np.random.RandomState(0)
    big = rng.uniform(-300, 300, (5000, 20))
    w = tuple(rng.uniform(0.1, 5.0, 20))
    assert np.array_equal(cr.state_cells(big, w).cpu().numpy(), np.floor(big / np.array(w)).astype(np.int32))
    # every episode of the example starts from (nearly) the same scene: with coarse cells it is ONE state in which all 11
    # actions were tried 15-16 times; the default grid splits it where a coordinate straddles a cell face
    ids, S = cr.index_states(obs)
    assert 1 <= S <= 40 and ids.shape == (170,)
    # the hand-written hash kernel against numpy.unique: same partition, ids by first appearance / by cell coordinates
    cells_h = cells.cpu().numpy()
    _, first_row, inv = np.unique(cells_h, axis=0, return_index=True, return_inverse=True)
    inv = inv.reshape(-1)
    assert S == len(first_row)
    assert np.array_equal(cr.index_states(obs, order="cells")[0].cpu().numpy(), inv)
    appear = np.argsort(np.argsort(first_row))                    # rank of each distinct cell by its first row
    assert np.array_equal(ids.cpu().numpy(), appear[inv])
    ids1, S1 = cr.index_states(obs, (50.0, 50.0, 20.0, 20.0, 7.0) * 4)
    assert S1 == 1 and int(ids1.max().item()) == 0
    for ids, S in ((ids, S), (ids1, S1)):
        _check_path(dc, cr, co, torch, ids, S, action, reward)


def _check_path(dc, cr, co, torch, ids, S, action, reward):
    table = cr.to_reference_table(ids, action, reward)
    tbl = dc.RecordTable.from_reference_table(table, S, 11, storage=torch.float64)
    tr = dc.ConfidenceEstimator().trace(tbl)
    order = np.argsort(ids.cpu().numpy(), kind="stable")
    lens = np.bincount(ids.cpu().numpy(), minlength=S)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ref = co.trace(reward[order], action[order].astype(np.uint8), off, S, 11)
    assert np.array_equal(tr.n.cpu().numpy(), ref["n"]) and np.array_equal(tr.amax.cpu().numpy(), ref["amax"])
    assert np.allclose(tr.V.cpu().numpy(), ref["V"], rtol=1e-10, atol=1e-10)
    # the biggest cell has seen every action more than n_thres = 10 times: its values are evaluated, and the arg-max is
    # not one of the colliding candidates (actions 2 and 9 end near -75)
    s_big = int(np.argmax(lens))
    if ref["n"][s_big].min() > 10:
        assert ref["amax"][s_big] not in (2, 9)
        assert S > 1 or np.all(ref["n"][0] >= 15)


@pytest.mark.gpu
@pytest.mark.parametrize("N,D,span,seed", [(1, 20, 3, 0), (1000, 1, 7, 1), (5000, 20, 1, 2), (200000, 20, 2, 3), (70000, 5, 40, 4),
                                           (3000, 64, 2, 5), (4096, 3, 1000000, 6)])
def test_state_ids_hash_kernel_vs_numpy_unique(N, D, span, seed):
    """dcarl_state_ids (sort-free: hash table + verify + prefix sum) against numpy.unique on random cell rows: from
    one state (span 1) over heavy duplication to all-distinct rows, D from 1 to the 64-coordinate limit."""
    from dcarl_amd import carla_records as cr
    rng = np.random.RandomState(seed)
    cells = rng.randint(-span, span, (N, D)).astype(np.int32)
    if N > 10:
        cells[N // 2:N // 2 + 5] = cells[:5]                      # guaranteed repeats far apart
    ids, n = cr.state_ids(cells)
    _, first_row, inv = np.unique(cells, axis=0, return_index=True, return_inverse=True)
    inv = inv.reshape(-1)
    assert n == len(first_row)
    appear = np.argsort(np.argsort(first_row))
    got = ids.cpu().numpy()
    assert np.array_equal(got, appear[inv])
    assert got[0] == 0 and got.max() == n - 1
    ids2, n2 = cr.state_ids(cells)                                # atomics inside, deterministic outside
    assert n2 == n and np.array_equal(ids2.cpu().numpy(), got)
    # a table sized for the distinct states (generous, exact, and too small: the overflow is detected and the call repeated)
    for hint in (4 * n, n, max(1, n // 8)):
        ids3, n3 = cr.state_ids(cells, max_states=hint)
        assert n3 == n and np.array_equal(ids3.cpu().numpy(), got), hint


@pytest.mark.gpu
@pytest.mark.parametrize("N,D", [(1, 20), (5000, 20), (123457, 20), (4000, 4), (3000, 64)])
def test_state_cells_with_hash_feeds_state_ids(N, D):
    """The row hashes made by the cells kernel (one pass over the observations) give the same ids as hashing the rows again."""
    from dcarl_amd import carla_records as cr
    rng = np.random.RandomState(N + D)
    centres = rng.randint(-50, 50, (max(1, N // 50), D)) + 0.5
    obs = centres[rng.randint(0, len(centres), N)] + (rng.rand(N, D) - 0.5) * 0.9
    w = [1.0] * D
    cells, hashes = cr.state_cells(obs, w, want_hash=True)
    assert np.array_equal(cells.cpu().numpy(), np.floor(obs).astype(np.int32))
    assert np.array_equal(cr.state_cells(obs, w).cpu().numpy(), cells.cpu().numpy())
    a, na = cr.state_ids(cells)
    b, nb = cr.state_ids(cells, hashes, max_states=len(centres))
    assert na == nb and torch.equal(a, b)
    c, nc = cr.index_states(obs, w, max_states=len(centres))
    assert nc == na and torch.equal(c, a.to(torch.int64))
    # the one-call form (dcarl_index_states_f64: the cells kernel enters its rows into the table itself), with and without a hint
    # — and with a hint that is too small (overflow detected, repeated with the safe size)
    for hint in (None, len(centres), 1):
        fused = cr.index_states_fused(obs, w, max_states=hint)
        if D % 4:
            assert fused is None
            continue
        fc, fi, fn = fused
        assert fn == na and torch.equal(fi, a) and torch.equal(fc, cells)
    first = np.unique(np.floor(obs).astype(np.int64), axis=0, return_index=True)[1]
    assert na == len(first)
