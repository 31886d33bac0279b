"""Per-record cost of the online kernel against the candidate count, on equal streams (65 536 states x 20 000 records):
    gpurun -- 'python tools/experiments/bench_actions.py'"""
import sys, torch, numpy as np
sys.path.insert(0, '.')
import dcarl_amd as dc
dc.require_gpu()
est = dc.ConfidenceEstimator()
for A in (tuple(int(x) for x in sys.argv[1:]) or (16, 11, 12, 13, 16, 11)):
    q = torch.rand(A) * 150 - 50
    tbl = dc.sampler.sample_state_records(q, 20000, seed=1, S=65536)
    out = est.trace(tbl)
    est.trace(tbl, out=out); est.trace(tbl, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        est.trace(tbl, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(A, dc._lib.last_kernel(), round(ms, 3), 'ms', round(ms * 1e9 / (65536 * 20000), 3), 'ps/record')
    del tbl, out
