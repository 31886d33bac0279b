"""bench legs: SURVEY 8(f) rows (RLS neighbour statistics, episode returns, state ids, Frenet candidates) and the drop-in script shape."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

from .core import *          # noqa: F401,F403  (the shared vocabulary of the legs: log, timed, roofline, result, the rank helpers ...)
from .core import STATE
from .online import run_trace_table


def run_dropin_a30(dc, args, rank, world):
    """What the Sim1 drop-in script itself runs: A = 30 candidates declared (S1:39 action_num), 11 ever sampled, float64
    record storage, on replicas of the bundled table.  (The host narrows the launch to the 12 candidates that can matter,
    ConfidenceEstimator._narrowed; DCARL_NO_NARROW=1 times the 32-slot one-wave kernel instead.)"""
    S = (args.states or 65536) // 64 * 64
    T = (args.records or 20000) // 4 * 4
    d = np.load(os.path.join(REPO, "Simulation_testing/Simulation_1/data_carla.npy"))[:T]
    dev = dc.require_gpu()
    base = dc.RecordTable.from_state_major(d[:, 3], d[:, 2].astype(np.int64), [T], 30, storage=torch.float64)
    # replicate the real stream into every lane of every slice: element (slice, quad, lane, j) <- base (quad, lane 0, j)
    W, nq = S // 64, T // 4
    R64 = base.R.view(nq, 64, 4)[:, 0, :][None, :, None, :].expand(W, nq, 64, 4).reshape(-1).contiguous()
    a8 = base.act.view(nq, 64, 4)[:, 0, :][None, :, None, :].expand(W, nq, 64, 4).reshape(-1).contiguous()
    if os.environ.get("DCARL_NO_NARROW"):
        base.max_action = None
    t64 = dc.RecordTable(S=S, A=30, R=R64, act=a8, lengths=torch.full((S,), T, dtype=torch.int32, device=dev),
                         slice_row_off=torch.arange(W + 1, dtype=torch.int64, device=dev) * T, n_records=S * T,
                         max_action=base.max_action)
    res, _ = run_trace_table(dc, t64, args, rank, world, "the Sim1 drop-in script's own shape: A = 30, f64 storage, the "
                             "bundled record stream replicated", "weak", S * world)
    return res


# ---- SURVEY 8(f) workloads -------------------------------------------------------------------------------------------
def run_rls(dc, args, rank, world):
    """the field confidence test.  Table = 209 600 visited rows (the length of the reference's
    visited_value.txt; the states file itself is a missing blob, so rows are synthetic with the field log's shape),
    queries = 1 024 decisions x (rule action + 7 candidates)."""
    N = args.records or 209_600
    B = args.states or 1024
    rng = np.random.RandomState(rank)
    proto = rng.uniform(-20, 20, (64, 20))
    dist = np.array(dc.rls.VISITED_STATE_DIST)
    st = proto[rng.randint(0, 64, N)] + rng.normal(0, 0.4, (N, 20)) * dist[:20]
    states = np.column_stack([st, rng.randint(0, 8, N).astype(np.float64)])
    rls = dc.rls.RLS(states, -rng.rand(N))
    obs = states[rng.randint(0, N, B), :20] + rng.normal(0, 0.3, (B, 20)) * dist[:20]
    q = torch.from_numpy(np.stack([dc.rls.RLS.state_with_action(obs, a) for a in range(8)], 1).reshape(-1, 21)).to(rls.device)
    Q = q.shape[0]
    box = [None, None]

    def step(e0, e1):
        if e0 is not None:
            e0.record()
        cnt, mean, var = rls.statistics(q)
        box[0], box[1] = cnt, rls.decide(cnt, mean, var, 7)
        if e1 is not None:
            e1.record()

    step(None, None)
    dt, kern_ms = timed(step, args.steps, args.warmup, world)
    cnt, act = box
    alg = (N * 22 + Q * 21 + Q * 3) * 8 + B * 4              # table, queries, statistics, decisions: each touched once
    return result("box tests/sec (visited row x query point)", "tests/s", float(N) * Q * world, dt, args.steps, args.warmup,
                  world, "weak", "f64",
                  dict(workload="8(f) rank 2: RLS neighbour statistics + z-test", visited_rows=N, decisions=B, queries=Q,
                       mean_visited=float(cnt.double().mean().item()), rl_actions_taken=int((act != 0).sum().item())),
                  rls_roofline(alg, kern_ms, float(N) * Q))


def rls_roofline(alg, kern_ms, tests):
    """The scan lives in the L2 (38 MB of compulsory traffic): its roofline is COMPARE ISSUE.  A box test is up to 42
    v_cmp_le_f64 (21 dimensions x two faces) on a 64-query wavefront; v_cmp_*_f64 costs 2.23-2.51 ns per wave-instruction and
    SIMD at 3-4 waves per SIMD (profiles/r03_ubench_issue.txt).  peak = every test paying all 42 compares on all 1 024 SIMDs;
    the kernel leaves a row at the first group of bounds no lane satisfies, so it can exceed that "peak" on easy tables."""
    t_cmp = 2.37e-9
    peak = 1024 * 64 / (42 * t_cmp)
    ach = tests / (kern_ms * 1e-3)
    r = roofline(alg, kern_ms, "rls_partial_kernel")
    r.update(bound="valu compare issue", achieved=ach, peak=peak, unit="box tests/s", frac=ach / peak,
             hbm_frac=alg / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
             note="peak = 1024 SIMDs x 64 lanes / (42 f64 compares x 2.37 ns); early exits let the kernel skip compares; the "
                  "compulsory HBM bytes are tiny (hbm_frac), the table is an L2 resident")
    return r


def run_episodes(dc, args, rank, world):
    """episode-return reduction.  E episodes of 60 ... 600 simulator steps (a CARLA junction episode
    at 10 Hz), per step (vx, vy) f64 + a flag byte in, the step reward out, per episode the return and AveSpeed; then the
    field back-up (RLS.add_data) over the same reward stream.  Two launches per pass; the roofline figure is the pair's."""
    E = args.states or 2 ** 19
    g = torch.Generator(device="cuda").manual_seed(7 + rank)
    lens = torch.randint(60, 601, (E,), generator=g, device="cuda", dtype=torch.int64)
    ep_off = torch.zeros(E + 1, dtype=torch.int64, device="cuda")
    torch.cumsum(lens, 0, out=ep_off[1:])
    N = int(ep_off[-1].item())
    vx = torch.rand(N, generator=g, device="cuda", dtype=torch.float64) * 12.0
    vy = torch.rand(N, generator=g, device="cuda", dtype=torch.float64) * 3.0 - 1.5
    flags = torch.zeros(N, dtype=torch.uint8, device="cuda")
    last = ep_off[1:] - 1
    kind = torch.randint(0, 4, (E,), generator=g, device="cuda")               # how the episode ends: collision / passed / stuck / time-out
    flags[last] = torch.tensor([1, 2, 4, 0], dtype=torch.uint8, device="cuda")[kind]
    done = (kind != 3).to(torch.uint8)
    step_r = torch.empty(N, dtype=torch.float64, device="cuda")
    ep_r, ave = torch.empty(E, dtype=torch.float64, device="cuda"), torch.empty(E, dtype=torch.float64, device="cuda")
    value = torch.empty(N, dtype=torch.float64, device="cuda")
    rec = torch.empty(N, dtype=torch.uint8, device="cuda")
    gp = torch.from_numpy(dc.episodes.gamma_powers(0.95, 10)).cuda()
    lib, P, chk = dc._lib.load(), dc._lib.ptr, dc._lib.check

    def step(e0, e1):
        if e0 is not None:
            e0.record()
        chk(lib.dcarl_episode_returns_f64(P(vx), P(vy), P(flags), P(ep_off), E, P(step_r), P(ep_r), P(ave), dc._lib.stream_ptr()),
            "dcarl_episode_returns_f64")
        chk(lib.dcarl_nstep_backup_f64(P(step_r), P(ep_off), P(done), E, P(gp), 10, P(value), P(rec), dc._lib.stream_ptr()),
            "dcarl_nstep_backup_f64")
        if e1 is not None:
            e1.record()

    step(None, None)
    dt, kern_ms = timed(step, args.steps, args.warmup, world)
    alg = N * (17 + 8) + N * (8 + 9) + E * (8 + 8 + 8 + 1 + 2 * 8)             # per step, per transition, per episode
    return result("simulator steps reduced + backed up per second", "steps/s", float(N) * world, dt, args.steps, args.warmup,
                  world, "weak", "f64",
                  dict(workload="8(f) rank 4: episode returns (TestScenario_Town03 reward) + n-step / gamma back-up (RLS.add_data)",
                       episodes=E, steps=N, mean_steps_per_episode=N / E),
                  roofline(alg, kern_ms, "episode_returns_kernel + nstep_backup_kernel"))


def run_state_ids(dc, args, rank, world):
    """observation rows -> grid cells -> dense state ids (hash kernels, no sort).  N records of 20-dim
    observations drawn around 2^17 prototype states (CARLA tables revisit states heavily), cell width 1."""
    N = args.records or 2 ** 24
    D, protos = 20, args.states or 2 ** 17
    g = torch.Generator(device="cuda").manual_seed(11 + rank)
    centre = torch.randint(-200, 200, (protos, D), generator=g, device="cuda").to(torch.float64) + 0.5
    which = torch.randint(0, protos, (N,), generator=g, device="cuda")
    obs = centre[which] + (torch.rand((N, D), generator=g, device="cuda", dtype=torch.float64) - 0.5) * 0.9
    del which
    lib, P, chk = dc._lib.load(), dc._lib.ptr, dc._lib.check
    cells = torch.empty((N, D), dtype=torch.int32, device="cuda")
    hashes = torch.empty(N, dtype=torch.int64, device="cuda")
    ids = torch.empty(N, dtype=torch.int32, device="cuda")
    out = torch.zeros(3, dtype=torch.int64, device="cuda")
    hint = 2 * protos                                         # the caller's estimate of the distinct states (CARLA tables revisit states)
    ws = torch.empty(int(lib.dcarl_workspace_bytes(3, hint, 0, N)), dtype=torch.uint8, device="cuda")
    width = torch.ones(D, dtype=torch.float64, device="cuda")

    def step(e0, e1):
        if e0 is not None:
            e0.record()
        if os.environ.get("DCARL_BENCH_STATE_IDS_TWO_CALLS") == "1":      # (A/B: the two-call form, hashes through HBM)
            chk(lib.dcarl_state_cells_f64(P(obs), N, D, P(width), P(cells), P(hashes), dc._lib.stream_ptr()), "dcarl_state_cells_f64")
            chk(lib.dcarl_state_ids(P(cells), P(hashes), N, D, hint, P(ws), P(ids), P(out), dc._lib.stream_ptr()), "dcarl_state_ids")
        else:
            chk(lib.dcarl_index_states_f64(P(obs), N, D, P(width), hint, P(ws), P(cells), P(ids), P(out), dc._lib.stream_ptr()),
                "dcarl_index_states_f64")
        if e1 is not None:
            e1.record()

    step(None, None)
    n_states, clashes, overflow = (int(v) for v in out.cpu())
    if overflow:
        raise RuntimeError("state_ids: the hash table sized for the distinct-state estimate overflowed")
    dt, kern_ms = timed(step, args.steps, args.warmup, world)
    alg = N * (8 * D + 4 * D) + N * (4 * D + 4)            # cells kernel: obs in, cells out; id kernels: cells in (once), ids out
    return result("records indexed per second", "records/s", float(N) * world, dt, args.steps, args.warmup, world, "weak", "i32",
                  dict(workload="8(f) rank 1: observation rows -> grid cells -> dense state ids", records=N, dims=D,
                       distinct_states=n_states, hash_clashes=clashes),
                  roofline(alg, kern_ms, "state_cells_hash_kernel<insert> + state_ids_{clear,verify,assign}_kernel + bit-word prefix",
                           note="dcarl_index_states_f64: the cells kernel hashes its rows and enters them into the id table itself, the "
                                "table is sized for the distinct-state estimate (2 x 2^17 slots of 16 B); the verify pass re-reads the "
                                "cell rows: algorithmic bytes count every array once"))


def run_frenet(dc, args, rank, world):
    """Frenet candidate generation (10 candidates x 14 samples x 8 fields per start state)."""
    B = args.states or 2 ** 20
    rng = np.random.RandomState(rank)
    fs = dc.frenet.FrenetSampler()
    start = torch.from_numpy(np.column_stack([rng.uniform(0, 500, B), rng.uniform(0, 15, B), rng.uniform(-4, 4, B),
                                              rng.uniform(-2, 2, B), np.zeros(B)])).to(fs.device)
    out = fs.calc_frenet_paths(start, None, None, None, None)

    def step(e0, e1):
        if e0 is not None:
            e0.record()
        fs.calc_frenet_paths(start, None, None, None, None, out=out)
        if e1 is not None:
            e1.record()

    dt, kern_ms = timed(step, args.steps, args.warmup, world)
    NC, NT = fs.n_candidates, fs.grid.nt_max
    alg = B * (5 * 8 + NC * 8 * NT * 8 + NC * 3 * 8)
    return result("candidate trajectories/sec", "candidates/s", float(B) * NC * world, dt, args.steps, args.warmup, world,
                  "weak", "f64",
                  dict(workload="8(f) rank 3: calc_frenet_paths, 10 candidates x 14 samples x 8 fields", start_states=B),
                  roofline(alg, kern_ms, "frenet_samples_kernel"))


def run_frenet_plan(dc, args, rank, world):
    """SURVEY 8(f) rank 3, whole chain: calc_frenet_paths -> calc_global_paths -> get_optimal_trajectory (4 obstacles)."""
    from dcarl_amd import frenet as fr
    B = args.states or 2 ** 19
    rng = np.random.RandomState(rank)
    fs = fr.FrenetSampler()
    wx = np.linspace(0.0, 900.0, 61)
    path = fr.ReferencePath(wx, 30.0 * np.sin(wx / 120.0), fs.device)
    start = torch.from_numpy(np.column_stack([rng.uniform(0, 800, B), rng.uniform(0, 12, B), rng.uniform(-3, 3, B),
                                              rng.uniform(-1, 1, B), np.zeros(B)])).to(fs.device)
    sx = start[:, 0].cpu().numpy()
    obs = np.stack([np.column_stack([sx + rng.uniform(5, 45, B), 30.0 * np.sin(sx / 120.0) + rng.uniform(-5, 5, B),
                                     rng.uniform(-2, 8, B), rng.uniform(-1, 1, B), rng.uniform(-1, 1, B)]) for _ in range(4)], 1)
    obs = torch.from_numpy(obs).to(fs.device)
    cands = fs.calc_frenet_paths(start, None, None, None, None)
    box = [None]

    def step(e0, e1):
        if e0 is not None:
            e0.record()
        fs.calc_frenet_paths(start, None, None, None, None, out=cands)
        gp = fr.calc_global_paths(fs, cands, path)
        box[0] = fr.get_optimal_trajectory(fs, cands, gp, obs)
        if e1 is not None:
            e1.record()

    step(None, None)
    dt, kern_ms = timed(step, args.steps, args.warmup, world)
    choice = box[0]
    NC, NT = fs.n_candidates, fs.grid.nt_max
    # candidates written once and read twice (global paths: d and s; selection: s_d, s_dd), global paths written + read
    alg = B * (5 * 8 + NC * 8 * NT * 8 + NC * 24 + NC * 2 * NT * 8 + NC * (5 * NT * 8 + 4) + NC * (2 * NT + 3 * NT) * 8 + 4 * 40 + 4)
    return result("planning decisions/sec", "decisions/s", float(B) * world, dt, args.steps, args.warmup, world, "weak", "f64",
                  dict(workload="8(f) rank 3: candidates + global paths + screening/selection, 4 obstacles", start_states=B,
                       brake_fraction=float((choice == 0).double().mean().item())),
                  roofline(alg, kern_ms, "frenet_samples_kernel + frenet_global_kernel + frenet_select_kernel"))
