"""Strong-scaling shards of configs[3] in online mode: slices per workgroup (DCARL_TRACE_SLICES = 1..4, read by the launcher at
every launch) against the default choice.  A shard of 8 holds 2 048 slices = two rounds of four-slice workgroups whose duration
is set by the longest stream (2 400 records); fewer slices per workgroup give the dispatcher a finer grain and spread a slice's
three waves over several SIMDs.  Prints ms per (world, slices) for shard 0 and the full table."""
import os, sys, importlib.util
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import dcarl_amd as dc
spec = importlib.util.spec_from_file_location("bench", os.path.join(REPO, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)

dc.require_gpu()
est = dc.ConfidenceEstimator()


def time_it(fn, n=40):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


total = int(os.environ.get("TOTAL", 2 ** 20))
for world in [int(w) for w in os.environ.get("WORLDS", "8,4,2,1").split(",")]:
    tbl, part, _ = bench.cfg3_shard(dc, total, world, 0, 1000.0, "balanced")
    o = est.trace(tbl)
    row = []
    for ns in ("", "4", "3", "2", "", "4", "3"):
        if ns:
            os.environ["DCARL_TRACE_SLICES"] = ns
        else:
            os.environ.pop("DCARL_TRACE_SLICES", None)
        row.append((ns or "default", round(time_it(lambda: est.trace(tbl, out=o)), 4)))
    os.environ.pop("DCARL_TRACE_SLICES", None)
    print(f"world {world}: states {tbl.S} slices {(tbl.S + 63) // 64} records {tbl.n_records:.3e}  " +
          "  ".join(f"ns={a}: {b} ms" for a, b in row), flush=True)
    tbl = o = None
    torch.cuda.empty_cache()
