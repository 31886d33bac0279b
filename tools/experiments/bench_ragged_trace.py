"""Online kernel on ragged slices: (a) lengths ~ Poisson around a slowly varying mean (configs[3]-like: most records in
the guard-free path), (b) lengths uniform in [0, 2T) (worst case: every slice contains short streams, all records go
through the per-lane guarded path)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dcarl_amd as dc

S, T, A = 65536, 1000, 11
dev = dc.require_gpu()
q = torch.from_numpy(np.random.RandomState(0).uniform(-50, 100, (S, A)).astype(np.float32))
dense = dc.sampler.sample_state_records(q, 2 * T, seed=1)
rng = np.random.RandomState(1)
u = rng.randint(0, 2 * T, S)
for name, lens in (("poisson(mean 1000*(0.5+s/S))", rng.poisson(T * (0.5 + np.arange(S) / S))),
                   ("uniform[0,2000)", u), ("uniform[0,2000) sorted slots", -np.sort(-u))):
    lens = np.minimum(lens, 2 * T).astype(np.int32)
    tbl = dc.RecordTable(S=S, A=A, R=dense.R, act=dense.act, lengths=torch.from_numpy(lens).to(dev),
                         slice_row_off=dense.slice_row_off, n_records=int(lens.sum()))
    est = dc.ConfidenceEstimator()
    out = est.trace(tbl)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        est.trace(tbl, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{name:32s} records {tbl.n_records:.3e}  {ms:.3f} ms  {tbl.n_records / ms / 1e6:.1f} Grec/s... ", f"{10 * tbl.n_records / ms / 1e6 / 8000 * 100:.1f} % of 8 TB/s")
