"""Where configs[3].batch's 1.19x HBM read traffic comes from: the final-state kernel on uniform CSR tables of one bucket length each
(and one random-length table), five launches per table; run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and parsed by
exp_bounds_traffic_parse.py (launches in the order printed here).
    python tools/experiments/exp_bounds_traffic.py"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dcarl_amd as dc

dc.require_gpu()
dev = torch.device("cuda:0")
S0, A = 1 << 18, 11
S = S0
est = dc.ConfidenceEstimator()
plan = []
g = torch.Generator(device="cpu").manual_seed(5)
for name, lens in (("n=91 (364 B)", torch.full((S * A,), 91)), ("n=96 (384 B)", torch.full((S * A,), 96)), ("n=64 (256 B)", torch.full((S * A,), 64)),
                   ("n=93", torch.full((S * A,), 93)), ("n=23", torch.full((S * A,), 23)), ("n=364", torch.full((S * A,), 364)), ("n=729", torch.full((S * A // 2,), 729)), ("n=1818", torch.full((S * A // 4,), 1818)),
                   ("n~U(1,181)", torch.randint(1, 182, (S0 * A,), generator=g))):
    S = lens.numel() // A
    seg = torch.zeros(S * A + 1, dtype=torch.int64)
    seg[1:] = torch.cumsum(lens.to(torch.int64), 0)
    n = int(seg[-1])
    vals = torch.randn(n, device=dev, dtype=torch.float32)
    seg = seg.to(dev)
    hint = max(1, n // (S * A))
    r = est.bounds(vals, S, A, seg_off=seg, n_mean_hint=hint)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(4):
        est.bounds(vals, S, A, seg_off=seg, n_mean_hint=hint, out=r)
    ev[1].record()
    torch.cuda.synchronize()
    plan.append(dict(name=name, launches=5, sample_bytes=4 * n, seg_bytes=8 * (S * A + 1), out_bytes=S * (12 * A + 8), ms=ev[0].elapsed_time(ev[1]) / 4,
                     kernel=dc._lib.last_kernel() if hasattr(dc._lib, "last_kernel") else None))
    del vals, seg, r
    torch.cuda.empty_cache()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(plan, open("gpurun_out/exp_bounds_traffic_plan.json", "w"), indent=1)
for p in plan:
    print(p)
