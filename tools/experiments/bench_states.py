"""Online kernel against the number of states (4 096 records per stream): how well small tables fill the GPU.
Each size twice: slices per workgroup chosen by the launcher, and pinned to 4 (DCARL_TRACE_SLICES=4, the only shape before).
    gpurun -- 'python tools/experiments/bench_states.py [A]'"""
import os, sys, torch
sys.path.insert(0, '.')
import dcarl_amd as dc
dc.require_gpu()
est = dc.ConfidenceEstimator()
A = int(sys.argv[1]) if len(sys.argv) > 1 else 11
T = 4096
for S in (64, 4096, 16384, 32768, 49152, 65536, 66560, 81920, 98304, 131072, 147456):
    q = torch.rand(A) * 150 - 50
    tbl = dc.sampler.sample_state_records(q, T, seed=1, S=S)
    out = est.trace(tbl)
    row = []
    for pin in (None, '4'):
        if pin: os.environ['DCARL_TRACE_SLICES'] = pin
        else: os.environ.pop('DCARL_TRACE_SLICES', None)
        est.trace(tbl, out=out); est.trace(tbl, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            est.trace(tbl, out=out)
        e1.record(); torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / 5)
    os.environ.pop('DCARL_TRACE_SLICES', None)
    print(A, S, 'auto %.3f ms  %.3f ps/record | pinned-4 %.3f ms  %.3f ps/record' % (row[0], row[0] * 1e9 / (S * T), row[1], row[1] * 1e9 / (S * T)), flush=True)
    del tbl, out
