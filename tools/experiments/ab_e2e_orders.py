"""configs[1] from the (N,4) f64 table under two arrival orders, both ingest paths, same box:
  dense   every state receives its t-th record before any receives its (t+1)-th (bench.py's table: RecordTable.to_reference_table)
  random  the same rows in a uniformly random order (torch.randperm): every state still holds exactly 20 000 records, but the states'
          progress spreads by +-sqrt(t) records
    gpurun -- 'python tools/experiments/ab_e2e_orders.py [states]'"""
import os, sys, torch
sys.path.insert(0, '.')
import dcarl_amd as dc
dc.require_gpu()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
tbl = dc.sampler.sample_state_records(dc.workloads.sim1_q_row(), 20000, seed=0, stream_id=0, S=S)
d = tbl.to_reference_table(dense_order=True)
del tbl
est = dc.ConfidenceEstimator()
for order in ("dense", "random"):
    if order == "random":
        g = torch.Generator(device='cuda').manual_seed(1)
        perm = torch.randperm(d.shape[0], generator=g, device='cuda')
        d = d[perm]
        del perm
        torch.cuda.empty_cache()
    ref = None
    for mode in ("0", "1", "0", "1"):
        os.environ["DCARL_INGEST_DIRECT"] = mode
        ts = []
        for _ in range(3):
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            t = dc.RecordTable.from_reference_table(d, S, 11, arrival=False)
            e1.record()
            tr = est.trace(t)
            e2.record()
            torch.cuda.synchronize()
            ts.append((e0.elapsed_time(e1), e1.elapsed_time(e2)))
            if ref is None: ref = (t.R.clone(), t.act.clone())
            ok = torch.equal(t.R, ref[0]) and torch.equal(t.act, ref[1])
            del t, tr
        print(order, "direct" if mode == "1" else "sort  ", "same table" if ok else "MISMATCH",
              " ".join(f"{a:.2f}+{b:.2f}={a + b:.2f}" for a, b in ts), flush=True)
    del ref
