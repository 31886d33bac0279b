"""Run-to-run determinism of the online kernel at full size: the same table through the same launch N times, every output compared bit
for bit with the first run's (a hand-over that ever let a wave read a stale row would show as a differing trace element).
    python tools/soak_trace_determinism.py [launches per table]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DCARL_BENCH_DEVICE", "cuda")
import dcarl_amd as dc
from bench_legs.core import build_trace_workload

dc.require_gpu()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
est = dc.ConfidenceEstimator()
tables = [("configs[1] 65 536 x 20 000, A = 11", lambda: build_trace_workload(dc, 65536, 20000, 0)),
          ("configs[3] 2^18 ragged states (Sim2 law), A = 11", lambda: dc.workloads.sim2_ragged(1 << 18, 0, 1 << 18)[0]),
          ("configs[4] 2^19 mixed states x 64 per bucket, A = 16", lambda: dc.workloads.mixed_records(1 << 19, n=64, seed=0, lo_state=0, stream_id=0)[0])]
bad = 0
for name, make in tables:
    tbl = make()
    out = est.trace(tbl)
    keep = [t.clone() for t in (out.step_val, out.step_act, out.V, out.n, out.vmax, out.amax, out.activation_step)]   # (the same buffers are
    # written again and again: what the kernel never writes — the layout's padding — stays what it was)
    t0 = time.time()
    diff = 0
    for i in range(N):
        est.trace(tbl, out=out)
        same = all(torch.equal(a, b) for a, b in zip(keep, (out.step_val, out.step_act, out.V, out.n, out.vmax, out.amax, out.activation_step)))
        diff += 0 if same else 1
    torch.cuda.synchronize()
    print(f"{name}: {N} launches, {diff} differing from the first  ({dc._lib.last_kernel()}, {tbl.n_records} records, {time.time() - t0:.1f} s)")
    bad += diff
    del tbl, keep, out
    torch.cuda.empty_cache()
print("determinism soak:", "all identical" if bad == 0 else f"{bad} DIFFERING LAUNCHES")
sys.exit(1 if bad else 0)
