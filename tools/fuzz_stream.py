"""Randomised cross-check of the streamed pipeline (dcarl_amd.stream.trace_stream) and of the pairs ingest (RecordTable.from_pairs)
against the one-piece device-resident pass (run on the GPU box):
    python tools/fuzz_stream.py [iterations] [seed]
Each iteration draws states, candidates, table size, an arrival law, the chunk size, storage, which outputs come back, the source
kind (array / strided view / iterable of ragged pieces) and compares the carried state and the per-arrival traces bit for bit."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import dcarl_amd as dc
from dcarl_amd.stream import trace_stream

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
est = dc.ConfidenceEstimator()
bad = 0
for it in range(iters):
    S = int(rng.choice([1, 3, 20, 64, 65, 300, 2048, 5000, 65536]))
    A = int(rng.choice([1, 2, 5, 11, 12, 16, 17, 30]))
    N = int(rng.choice([1, 7, 1000, 6656, 6657, 50_000, 300_000, (1 << 20) + 3, 3_000_000]))
    law = rng.choice(["uniform", "skewed", "runs"])
    if law == "uniform":
        st = rng.randint(0, S, N)
    elif law == "skewed":
        st = np.minimum((rng.exponential(S / 6.0, N)).astype(np.int64), S - 1)
    else:
        st = np.repeat(rng.randint(0, S, N // 16 + 1), 16)[:N]
    data = np.zeros((N, 4))
    data[:, 0], data[:, 1], data[:, 2] = st, rng.rand(N), rng.randint(0, A, N)
    data[:, 3] = (rng.randn(N) * 50 + 20).astype(np.float32)
    storage = torch.float32 if rng.rand() < 0.6 else torch.float64
    steps = bool(rng.rand() < 0.5)
    overall = bool(steps and rng.rand() < 0.5)
    chunk = int(rng.choice([1, 5, 333, 4096, 6656, 100_000, 1 << 20, (1 << 20) + 1, 1 << 22]))
    if N // chunk > 400:
        chunk = N // 400 + 1
    one = est.trace(dc.RecordTable.from_reference_table(data, S, A, storage=storage, arrival=steps), want_steps=steps).check()
    kind = rng.choice(["array", "view", "pieces"])
    if kind == "array":
        src = data
    elif kind == "view":
        wide = np.zeros((N, 6))
        wide[:, 1:5] = data
        src = wide[:, 1:5]                                  # strided rows: goes through the staging buffers
    else:
        cuts = np.sort(rng.randint(0, N + 1, 4))
        src = iter([data[a:b] for a, b in zip([0, *cuts], [*cuts, N])])
    r = trace_stream(src, S, A, chunk_records=chunk, storage=storage, want_steps=steps, with_overall=overall, est=est,
                     copy_threads=int(rng.choice([1, 3, 8])))
    ok = bool(torch.equal(r.state.V, one.V) and torch.equal(r.state.n, one.n) and torch.equal(r.state.act_step, one.activation_step))
    if steps:
        sv, sa = one.steps_in_arrival_order()
        ok = ok and np.array_equal(r.step_val, sv.cpu().numpy()) and np.array_equal(r.step_act, sa.cpu().numpy())
    if overall:
        ov = est.overall_value(one).cpu().numpy()
        ok = ok and float(np.max(np.abs(r.overall_value - ov) / np.maximum(np.abs(ov), 1.0))) <= 1e-12
    # the same records as the sampler's arrays (with a few dropped visits mixed in) through the pairs ingest
    if storage == torch.float32:
        drop = rng.rand(N) < 0.01
        idx = np.where(drop, -1, st).astype(np.int32)
        t = dc.RecordTable.from_pairs(idx, data[:, 2].astype(np.int32), data[:, 3].astype(np.float32), S, A)
        ref = dc.RecordTable.from_reference_table(data[~drop], S, A, storage=torch.float32, arrival=False)
        ok = ok and bool(torch.equal(t.R, ref.R) and torch.equal(t.act, ref.act) and torch.equal(t.lengths, ref.lengths))
        b = est.bounds_from_table(t)
        full = est.trace(ref, want_steps=False)
        ok = ok and bool(torch.equal(b.V, full.V) and torch.equal(b.amax, full.amax) and torch.equal(b.n, full.n))
    bad += not ok
    print(f"{it:3d} S={S:6d} A={A:2d} N={N:8d} {law:8s} {'f32' if storage == torch.float32 else 'f64'} chunk={chunk:8d} {kind:6s} "
          f"steps={int(steps)} overall={int(overall)} chunks={r.chunks:4d} {'ok' if ok else 'MISMATCH'}", flush=True)
print("all ok" if not bad else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
