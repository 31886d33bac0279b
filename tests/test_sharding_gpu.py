"""Sharded result == unsharded result: the W shards of a table are run ONE AFTER THE OTHER on the one
GPU, their per-state summaries are reassembled exactly as the all-gather would deliver them (dist.assemble_summaries -> the same
SummaryTable code path), and the table must equal the single-table run bit for bit — arg-max, the f32 max's bit pattern, the
activation step — state by state.  configs[3] (2^20 ragged states, both modes, both partitions, world 2 / 4 / 8) and
configs[4] (the full 2^22-state table in 8 pieces)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dc():
    import dcarl_amd
    dcarl_amd.require_gpu()
    return dcarl_amd


def summaries(dc, est, tbl, mode):
    if mode == "trace":
        tr = est.trace(tbl, want_steps=False)
        return tr.amax.clone(), tr.vmax.clone(), tr.activation_step.clone()
    vals, seg = tbl.to_buckets()
    b = est.bounds(vals, tbl.S, tbl.A, seg_off=seg)
    return b.amax.clone(), b.vmax.clone(), None


@pytest.fixture(scope="module")
def cfg3_full(dc):
    """The whole configs[3] table on one GPU (states in state order), its summaries in both modes."""
    total = 1 << 20
    lengths = dc.workloads.sim2_visit_lengths(total, mean=1000.0, seed=0)
    tbl, _ = dc.workloads.sim2_table(total, torch.arange(total), A=11, mean=1000.0, seed=0, lengths_all=lengths)
    est = dc.ConfidenceEstimator()
    out = {mode: summaries(dc, est, tbl, mode) for mode in ("trace", "batch")}
    assert torch.equal(out["trace"][0], out["batch"][0]) and torch.equal(out["trace"][1], out["batch"][1])   # online table == batch table
    n = tbl.n_records
    del tbl
    torch.cuda.empty_cache()
    return total, lengths, out, n


@pytest.mark.parametrize("world,kind", [(8, "balanced"), (8, "contiguous"), (4, "balanced"), (2, "balanced")])
@pytest.mark.parametrize("mode", ["trace", "batch"])
def test_configs3_shards_equal_the_single_table(dc, cfg3_full, world, kind, mode):
    total, lengths, full, n_full = cfg3_full
    part = dc.layout.StatePartition.balanced(lengths, world) if kind == "balanced" else dc.layout.StatePartition.contiguous(total, world)
    est = dc.ConfidenceEstimator()
    blocks, recs = [], []
    for q in range(world):
        tbl, _ = dc.workloads.sim2_table(total, part.states_of(q), A=11, mean=1000.0, seed=0, lengths_all=lengths,
                                         sort_by_length=(kind != "balanced"))
        assert tbl.S == part.count(q)
        recs.append(tbl.n_records)
        blocks.append(summaries(dc, est, tbl, mode))
        del tbl
    assert sum(recs) == n_full
    if kind == "balanced":
        assert max(recs) / (sum(recs) / world) <= 1.02                              # VERDICT r3 item 1(i) on the real table
    elif world == 8:
        assert max(recs) / (sum(recs) / world) > 2.0                                # the old scheme: 27 % of the records on one rank
    a, v, s = dc.dist.assemble_summaries(part, blocks).states()
    fa, fv, fs = full[mode]
    assert torch.equal(a, fa)                                                       # arg-max, state by state
    assert torch.equal(v.view(torch.int32), fv.view(torch.int32))                   # max V: the f32 BIT PATTERN
    if mode == "trace":
        assert torch.equal(s, fs)                                                   # activation step
        assert int((s >= 0).sum()) > total // 4
    else:
        assert bool((s == -1).all())                                                # final-state mode has no latch: "never"


def test_configs4_full_table_in_8_pieces_equals_one_piece(dc):
    """configs[4] at its FULL stated size — 2^22 states x 16 candidates, even states the Sim1 row (11 live + 5 empty), odd states
    16 live, 64 samples per live bucket: 3.6e9 records, 18 GB of inputs — run as one table on one GPU and as the 8 contiguous
    shards an 8-GPU node would hold (every state holds 704 or 1 024 records: equal-state blocks ARE balanced here)."""
    total, world = 1 << 22, 8
    est = dc.ConfidenceEstimator()
    tbl, _, _ = dc.workloads.mixed_records(total, n=64, seed=0, lo_state=0)
    assert tbl.n_records == (total // 2) * (11 + 16) * 64
    fa, fv, fs = summaries(dc, est, tbl, "trace")
    del tbl
    torch.cuda.empty_cache()
    part = dc.layout.StatePartition.contiguous(total, world)
    blocks = []
    for q in range(world):
        lo, hi = dc.layout.shard_states(total, world, q)
        t, _, _ = dc.workloads.mixed_records(hi - lo, n=64, seed=0, lo_state=lo)
        blocks.append(summaries(dc, est, t, "trace"))
        del t
    a, v, s = dc.dist.assemble_summaries(part, blocks).states()
    assert torch.equal(a, fa) and torch.equal(v.view(torch.int32), fv.view(torch.int32)) and torch.equal(s, fs)
    even = torch.arange(total, device=a.device) % 2 == 0
    assert int(a[even].max()) <= 10 and int(a[~even].max()) == 15                   # the 5 empty candidates of even states never win


def test_world2_shards_through_a_real_gather_on_one_gpu(dc):
    """The gather object itself (world 1 here: no process group) with a balanced partition of a small ragged table: zero-copy
    slots as kernel outputs, reassembly in state order."""
    total = 4096
    lengths = dc.workloads.sim2_visit_lengths(total, mean=200.0, seed=1)
    part = dc.layout.StatePartition.balanced(lengths, 1)
    tbl, _ = dc.workloads.sim2_table(total, part.states_of(0), A=11, mean=200.0, seed=1, lengths_all=lengths, sort_by_length=False)
    est = dc.ConfidenceEstimator()
    ref = est.trace(tbl, want_steps=False)
    g = dc.dist.SummaryGather(total, tbl.device, part=part)
    slot = g.slot(0)
    out = est.trace(tbl, want_steps=False)
    out.amax, out.vmax, out.activation_step = slot.amax, slot.vmax, slot.act_step
    est.trace(tbl, want_steps=False, out=out)
    a, v, s = g.post(slot).states()
    order = part.states_of(0).to(a.device)
    assert torch.equal(a[order], ref.amax) and torch.equal(v[order], ref.vmax) and torch.equal(s[order], ref.activation_step)
