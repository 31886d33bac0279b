"""What the release / acquire fences around the LDS hand-over counters cost (the default; DCARL_TRACE_FENCED=0 = the bare
hardware-ordering form, trace_nwave_impl.h):
the headline table and the configs[4] shard shape, interleaved repeats.   gpurun -- 'python tools/experiments/ab_fenced.py'"""
import os, sys, torch
sys.path.insert(0, '.')
import dcarl_amd as dc
dc.require_gpu()
est = dc.ConfidenceEstimator()
for A, S, T in ((11, 65536, 20000), (16, 65536, 1000)):
    q = torch.linspace(-50, 100, A)
    tbl = dc.sampler.sample_state_records(q, T, seed=0, S=S)
    out = est.trace(tbl)
    res = {}
    for rep in range(3):
        for mode in ("plain", "fenced"):
            if mode == "plain": os.environ["DCARL_TRACE_FENCED"] = "0"
            else: os.environ.pop("DCARL_TRACE_FENCED", None)
            est.trace(tbl, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): est.trace(tbl, out=out)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(mode, []).append(e0.elapsed_time(e1) / 5)
    print(A, S, T, {k: ['%.3f' % x for x in v] for k, v in res.items()}, dc._lib.last_kernel(), flush=True)
