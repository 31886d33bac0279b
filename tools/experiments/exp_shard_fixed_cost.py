"""Final-state kernel on balanced shards of configs[3] of every size (world = 64 ... 1): t(world) = t0 + T / world ?  The fixed part t0 is what
a strong-scaled launch cannot shed (launch ramp, first dependent loads, the drain of the last blocks); printed with a least-squares fit."""
import os, sys, importlib.util
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import dcarl_amd as dc
from bench_legs import sharded
dc.require_gpu()
est = dc.ConfidenceEstimator()
rows = []
for mode in ("batch", "trace"):
    for world in (64, 32, 16, 8, 4, 2, 1):
        fn, _, S, n = sharded.shard_step(dc, est, "cfg3", mode, world, 0)
        for _ in range(30):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        N = 200 if world >= 8 else 60
        e0.record()
        for _ in range(N):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / N
        rows.append((mode, world, S, n, ms))
        print(f"{mode} world {world:3d}: states {S:8d} records {n:11d}  {ms:.4f} ms   x world = {ms * world:.4f}", flush=True)
        fn = None
        torch.cuda.empty_cache()
    r = [(1.0 / w, ms) for m, w, _, _, ms in rows if m == mode]
    A = np.array([[1.0, x] for x, _ in r]); y = np.array([t for _, t in r])
    t0, T = np.linalg.lstsq(A, y, rcond=None)[0]
    print(f"{mode}: fit t = {t0 * 1e3:.1f} us + {T:.4f} ms / world", flush=True)
