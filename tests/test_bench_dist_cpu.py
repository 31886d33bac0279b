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
           "--warmup", "2", "--total-states", "1000", "--partition", partition]
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


def test_bench_self_launch_without_torchrun_environment():
    """`python bench.py --gpus 2` with no RANK / WORLD_SIZE in the environment starts the ranks itself."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(DCARL_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--workload", "stub", "--steps", "3", "--warmup", "1",
                          "--states", "130"], env=env, capture_output=True, text=True, timeout=300, cwd=REPO)
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
