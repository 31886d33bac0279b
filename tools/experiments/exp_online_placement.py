"""Is the online kernel's time sensitive to WHERE its four big arrays lie (as the sampler's was: profiles/r05_sampler_variance.txt)?
The configs[1] table is rebuilt behind dummy allocations of different sizes; 12 launches each, median kernel ms.
    python tools/experiments/exp_online_placement.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dcarl_amd as dc
from bench_legs.core import build_trace_workload

dc.require_gpu()
est = dc.ConfidenceEstimator()
res = []
for k, pad_mb in enumerate((0, 1, 3, 17, 64 + 5, 256 + 33, 1024 + 7, 2048 + 129)):
    torch.cuda.empty_cache()
    pad = torch.empty(int(pad_mb * (1 << 20) + k * 4096 * 3), dtype=torch.uint8, device="cuda") if pad_mb else None
    tbl = build_trace_workload(dc, 65536, 20000, 0)
    out = est.trace(tbl)
    ts = []
    for _ in range(14):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); est.trace(tbl, out=out); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts = ts[2:]
    res.append((pad_mb, float(np.median(ts)), min(ts), max(ts), tbl.R.data_ptr() % (1 << 21), out.step_val.data_ptr() % (1 << 21) if out.step_val is not None else -1))
    print(f"pad {pad_mb:5d} MB: median {np.median(ts):.3f}  min {min(ts):.3f}  max {max(ts):.3f} ms   R @ {tbl.R.data_ptr():#x}  step_val @ {out.step_val.data_ptr():#x}  {dc._lib.last_kernel()}")
    del tbl, out, pad
