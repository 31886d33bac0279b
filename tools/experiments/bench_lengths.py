"""Per-record cost of the online kernel against the stream length, on equal streams with the total record count fixed
(S x T = 2^29): separates the per-workgroup set-up / drain from the steady state.
    gpurun -- 'python tools/experiments/bench_lengths.py [A]'"""
import sys, torch
sys.path.insert(0, '.')
import dcarl_amd as dc
dc.require_gpu()
est = dc.ConfidenceEstimator()
A = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for T in (256, 704, 1024, 2048, 4096, 16384, 1024, 704):
    S = (1 << 29) // T // 256 * 256
    q = torch.rand(A) * 150 - 50
    tbl = dc.sampler.sample_state_records(q, T, seed=1, S=S)
    out = est.trace(tbl)
    est.trace(tbl, out=out); est.trace(tbl, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        est.trace(tbl, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(A, T, S, dc._lib.last_kernel(), round(ms, 3), 'ms', round(ms * 1e9 / (S * T), 3), 'ps/record', flush=True)
    del tbl, out
