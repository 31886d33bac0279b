"""bench.py's distributed control flow at world 2 and 4 on CPU ranks (gloo): init_dist, state sharding, SummaryGather slots
with the asynchronous all-gather under the next step, max / sum over ranks, the JSON line — with a stub in the kernel's place
(the first real 8-GPU run must not be the first time this code executes with world > 1)."""
import json
import os
import socket
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,partition", [(2, "balanced"), (4, "balanced"), (2, "contiguous")])
def test_bench_control_flow_under_torchrun(world, partition):
    """``balanced``: the states are dealt to the ranks as length-sorted slices (layout.StatePartition.balanced, what configs[3]
    runs with); every rank checks every step's gathered table state by state after the reassembly through the partition's map."""
    port = free_port()
    env = dict(os.environ, DCARL_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", str(world), "--workload", "stub", "--steps", "4",
           "--warmup", "2", "--total-states", "1000", "--partition", partition, "--strong-states3", "5000", "--strong-states4", "3000"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=REPO)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    r = json.loads(lines[0])
    assert r["n_gpus"] == world and r["steps"] == 4 and r["warmup"] == 2 and r["scaling"] == "strong"
    assert r["config"]["states_total"] == 1000 and r["config"]["backend"] == "gloo" and r["config"]["partition"] == partition
    assert r["config"]["tables_checked"] == 4 + 2 - 1        # every step's gathered table but the last was checked on every rank
    assert abs(r["value"] * r["ms_per_step"] * 1e-3 - 1000) < 1e-6 * 1000      # value = states of ALL ranks / step time
    assert r["roofline"]["kernel"] == "stub" and r["cpu_baseline"] is None
    # N > 1 also carries the strong-scaling legs: the full table on rank 0 alone, then the shards of all ranks
    # with the double-buffered all-gather, measured in the same invocation
    legs = {k: v for k, v in r["other_configs"].items() if ".strong." in k}
    assert set(legs) == {"stub.strong.balanced", "stub.strong.contiguous"}
    for k, leg in legs.items():
        assert "error" not in leg, leg
        for key in STRONG_KEYS:
            assert key in leg, (k, key)
        assert leg["gather_verified"] is True and leg["world"] == world and leg["partition"] == k.split(".")[-1]
        assert leg["ms_full_1gpu"] > 0 and leg["ms_sharded_max_rank"] > 0 and leg["gather_ms"] > 0
        assert abs(leg["speedup"] - leg["ms_full_1gpu"] / leg["ms_sharded_max_rank"]) < 1e-9
        assert 1.0 <= leg["records_max_over_mean"] < 1.2
    assert "strong_scaling_incomplete" not in r


STRONG_KEYS = ("ms_full_1gpu", "ms_sharded_max_rank", "gather_ms", "speedup", "records_max_over_mean", "partition", "transport",
               "gather_verified", "kernel_ms_full_1gpu", "kernel_ms_sharded_max_rank", "states_total", "world")


def test_bench_prints_the_headline_when_a_strong_leg_hangs():
    """A rank that never arrives in a strong-scaling leg (the collectives of the others wait for it): after --strong-deadline the
    headline line is printed with the legs collected so far and every rank leaves."""
    port = free_port()
    env = dict(os.environ, DCARL_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", DCARL_BENCH_TEST_HANG="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--workload", "stub", "--steps", "3",
           "--warmup", "1", "--total-states", "640", "--strong-deadline", "6"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=REPO)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-4000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["value"] > 0 and "exceeded" in r["strong_scaling_incomplete"]
    assert out.returncode == 0, out.stderr[-4000:]


def test_bench_self_launch_without_torchrun_environment():
    """`python bench.py --gpus 2` with no RANK / WORLD_SIZE in the environment starts the ranks itself."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(DCARL_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--workload", "stub", "--steps", "3", "--warmup", "1",
                          "--states", "130", "--strong-states3", "2000", "--strong-states4", "1000"], env=env, capture_output=True, text=True, timeout=300, cwd=REPO)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["n_gpus"] == 2 and r["config"]["states_total"] == 260


def test_bench_stub_single_process():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(DCARL_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--workload", "stub", "--steps", "2", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=300, cwd=REPO)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["n_gpus"] == 1 and r["config"]["collective"] == "none"
