"""Chip-filling cross-check of the online kernels against the C oracle: six tables of 33 000-100 000 states (all 256 CUs busy,
several rounds of workgroups, ragged / sorted / uniform lengths), every output compared.  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dcarl_amd as dc
from oracle import c_oracle as co
rng = np.random.RandomState(5)
est = dc.ConfidenceEstimator()
for (S, A, T, kind) in [(65536, 11, 400, "uniform"), (65600, 11, 203, "ragged"), (100000, 12, 150, "sorted"), (70000, 5, 333, "ragged"), (65536, 16, 128, "uniform"), (33000, 11, 1000, "sorted")]:
    lens = np.full(S, T) if kind == "uniform" else (rng.randint(0, T + 1, S) if kind == "ragged" else np.sort(rng.randint(max(T - 60, 0), T + 1, S))[::-1].copy())
    N = int(lens.sum()); act = rng.randint(0, A, N).astype(np.uint8); st = np.repeat(np.arange(S), lens)
    q = rng.uniform(-50, 100, (S, A)).astype(np.float32)
    R = (q[st, act] + 50.0 * rng.standard_normal(N).astype(np.float32)).astype(np.float32)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    tr = est.trace(dc.RecordTable.from_state_major(R, act, lens, A))
    ref = co.trace(R, act, off, S, A)
    sv, sa = tr.steps_by_state()
    ok = (np.array_equal(sa.cpu().numpy(), ref["step_act"]) and np.array_equal(tr.n.cpu().numpy(), ref["n"]) and np.array_equal(tr.amax.cpu().numpy(), ref["amax"])
          and np.array_equal(tr.activation_step.cpu().numpy(), ref["activation_step"]) and np.allclose(tr.V.cpu().numpy(), ref["V"], rtol=1e-10, atol=1e-10))
    print(S, A, T, kind, N, "ok" if ok else "MISMATCH", flush=True)
    assert ok
print("all ok")
