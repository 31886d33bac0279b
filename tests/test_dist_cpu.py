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
