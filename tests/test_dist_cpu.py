"""N>1 path on CPU: world_size-2 gloo processes exercising the state sharding and the summary all-gather."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import torch

from dcarl_amd import dist as ddist
from dcarl_amd import layout

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pack_unpack_roundtrip():
    amax = torch.tensor([0, 5, 10], dtype=torch.int32)
    vmax = torch.tensor([100.0, -1.5, 62.0959], dtype=torch.float32)
    step = torch.tensor([-1, 4438, 7], dtype=torch.int32)
    a, v, s = ddist.unpack_summary(ddist.pack_summary(amax, vmax, step))
    assert torch.equal(a, amax) and torch.equal(v, vmax) and torch.equal(s, step)
    a, v, s = ddist.allgather_summary(3, amax, vmax, step)          # world size 1: identity
    assert torch.equal(a, amax) and torch.equal(v, vmax) and torch.equal(s, step)


def test_combine_global_statistics():
    """The host side of the 272-byte-per-rank statistics: rows of dcarl_summary_t -> totals (no GPU needed)."""
    rows = torch.zeros((3, ddist.SUMMARY_WORDS), dtype=torch.int64)
    rows[:, 0] = torch.tensor([5, 0, 7])
    rows[:, 1] = torch.tensor([1.5, -2.25, 100.0], dtype=torch.float64).view(torch.int64)
    rows[0, 2:5] = torch.tensor([1, 2, 3]); rows[2, 2:5] = torch.tensor([10, 0, 1])
    g = ddist.combine_stats(rows, 3)
    assert g == dict(activated=12, sum_vmax=99.25, policy_hist=[11, 2, 4])


def test_balanced_partition_of_the_configs3_law():
    """under the Sim2 visit law at S = 2^20 (configs[3]) the equal-state contiguous blocks give the
    busiest of 8 ranks 27 % of the records (a 3.65x ceiling); the dealt length-sorted slices give every rank the same share
    within 2 %, every state is owned exactly once, and a rank's local order is sorted by length."""
    from dcarl_amd import workloads
    S = 1 << 20
    lengths = workloads.sim2_visit_lengths(S, mean=1000.0, seed=0, device="cpu")
    assert abs(float(lengths.sum()) / S - 1000.0) < 2.0 and int(lengths.max()) > 2000 and int(lengths.min()) < 60
    for world in (2, 4, 8):
        part = layout.StatePartition.balanced(lengths, world)
        owned = torch.zeros(S, dtype=torch.int32)
        share = []
        for q in range(world):
            st = part.states_of(q)
            assert st.numel() == part.count(q) <= part.per
            owned[st] += 1
            ln = lengths[st]
            assert bool((ln[:-1] >= ln[1:]).all())                         # no slot sort needed on the rank
            share.append(int(ln.sum()))
        assert bool((owned == 1).all())
        assert max(share) / (sum(share) / world) <= 1.02, (world, share)
        assert max(share) - min(share) <= 64 * int(lengths.max())
        g = part.global_index()
        assert g.shape == (world, part.per) and int((g >= 0).sum()) == S
        # the old scheme, for the record: its ceiling
        old = [int(lengths[slice(*layout.shard_states(S, world, q))].sum()) for q in range(world)]
        ceiling = sum(old) / max(old)
        assert ceiling < {2: 2.01, 4: 2.4, 8: 3.8}[world]
    assert 3.5 < ceiling < 3.8                                              # 8 ranks: 3.65x


def test_summary_table_reassembly_through_the_partition():
    rng = np.random.RandomState(3)
    for S, world in ((1, 2), (63, 2), (64, 4), (130, 3), (1000, 8), (5000, 4)):
        lengths = torch.from_numpy(rng.randint(0, 700, S).astype(np.int64))
        amax = torch.from_numpy(rng.randint(0, 11, S).astype(np.int32))
        vmax = torch.from_numpy(rng.uniform(-50, 100, S).astype(np.float32))
        step = torch.from_numpy(rng.randint(-1, 20000, S).astype(np.int32))
        for part in (layout.StatePartition.balanced(lengths, world), layout.StatePartition.contiguous(S, world)):
            blocks = [(amax[part.states_of(q)], vmax[part.states_of(q)], step[part.states_of(q)]) for q in range(world)]
            a, v, s = ddist.assemble_summaries(part, blocks).states()
            assert torch.equal(a, amax) and torch.equal(v, vmax) and torch.equal(s, step), (S, world, part.kind)


WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, os.environ["DCARL_REPO"])
    from dcarl_amd import dist as ddist, layout
    dist.init_process_group("gloo")
    w, r = dist.get_world_size(), dist.get_rank()
    for S in (1, 64, 130, 1000):
        rng = np.random.RandomState(S)
        amax = torch.from_numpy(rng.randint(0, 11, S).astype(np.int32))
        vmax = torch.from_numpy(rng.uniform(-50, 100, S).astype(np.float32))
        step = torch.from_numpy(rng.randint(-1, 20000, S).astype(np.int32))
        lo, hi = ddist.my_states(S)
        assert (lo, hi) == layout.shard_states(S, w, r)
        a, v, s = ddist.allgather_summary(S, amax[lo:hi], vmax[lo:hi], step[lo:hi])
        assert torch.equal(a, amax) and torch.equal(v, vmax) and torch.equal(s, step), (S, r)
        g = ddist.SummaryGather(S, "cpu")                 # the pre-allocated variant bench.py uses
        tab = g(amax[lo:hi], vmax[lo:hi], step[lo:hi])
        for q in range(w):
            qlo, qhi = layout.shard_states(S, w, q)
            ba, bv, bs = tab.block(q)
            assert torch.equal(ba, amax[qlo:qhi]) and torch.equal(bs, step[qlo:qhi]) and torch.equal(bv, vmax[qlo:qhi])
        a2, v2, s2 = tab.states()
        assert torch.equal(a2, amax) and torch.equal(v2, vmax) and torch.equal(s2, step)
        # zero-copy form (bench.py's timed loop): the "kernel" writes its per-state outputs INTO the slot's arrays, the
        # collective is posted from there; three steps in flight over two buffer sets, each table complete after wait()
        tabs = []
        g = ddist.SummaryGather(S, "cpu")
        for k in range(3):
            slot = g.slot(k)
            assert slot.amax.numel() == hi - lo and (hi == lo or slot.amax.data_ptr() == slot.buf.data_ptr())
            slot.amax.copy_((amax[lo:hi] + k) % 11)       # stands in for the kernel epilogue
            slot.vmax.copy_(vmax[lo:hi] + k)
            if k == 0:
                assert bool((slot.act_step == -1).all())  # kernels without a latch leave "never" there
            slot.act_step.copy_(step[lo:hi])
            tabs.append((k, g.post(slot, async_op=True)))
            if k >= 1:                                    # the table of step k-1 is still intact after step k was issued
                g.wait()
                kk, t = tabs[k - 1]
                for q in range(w):
                    qlo, qhi = layout.shard_states(S, w, q)
                    ba, bv, bs = t.block(q)
                    assert torch.equal(ba, (amax[qlo:qhi] + kk) % 11), (S, r, kk)
                    assert torch.equal(bv, vmax[qlo:qhi] + kk) and torch.equal(bs, step[qlo:qhi])
        g.wait()
        # the record-balanced partition (ragged tables: length-sorted slices dealt round-robin): every rank passes ITS states in
        # ITS local order, the gathered table comes back in state order through the partition's map
        lengths = torch.from_numpy(rng.randint(0, 500, S).astype(np.int64))
        part = ddist.partition(S, lengths)
        assert part.kind == "balanced" and part.world == w
        mine = part.states_of(r)
        a, v, s = ddist.allgather_summary(S, amax[mine], vmax[mine], step[mine], part=part)
        assert torch.equal(a, amax) and torch.equal(v, vmax) and torch.equal(s, step), ("balanced", S, r)
        g = ddist.SummaryGather(S, "cpu", part=part)
        assert g.n_local == mine.numel()
        slot = g.slot(0)
        slot.amax.copy_(amax[mine]); slot.vmax.copy_(vmax[mine]); slot.act_step.copy_(step[mine])
        tab = g.post(slot, async_op=True)
        g.wait()
        a, v, s = tab.states()
        assert torch.equal(a, amax) and torch.equal(v, vmax) and torch.equal(s, step), ("balanced slots", S, r)
        for q in range(w):
            ba, bv, bs = tab.block(q)
            qs = part.states_of(q)
            assert torch.equal(ba, amax[qs]) and torch.equal(bv, vmax[qs]) and torch.equal(bs, step[qs])
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(os.environ["DCARL_OUT"], f"rank{r}.ok"), "w").write("ok")   # (stdout of the ranks interleaves)
""")


def test_allgather_summary_two_ranks(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, DCARL_REPO=REPO, DCARL_OUT=str(tmp_path), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists()
