"""configs[1] end to end from the arrival-ordered (N,4) f64 table, per stage, both ingest paths (A/B on one box):
    gpurun -- 'python tools/experiments/bench_e2e_quick.py [states]'"""
import os, sys, time, torch
sys.path.insert(0, '.')
import dcarl_amd as dc
dc.require_gpu()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = 20000
q = dc.workloads.sim1_q_row()
tbl = dc.sampler.sample_state_records(q, T, seed=0, stream_id=0, S=S)
d = tbl.to_reference_table(dense_order=True)
ref_R, ref_act = tbl.R.clone(), tbl.act.clone()
del tbl
est = dc.ConfidenceEstimator()
for mode in ("0", "1", "0", "1"):
    os.environ["DCARL_INGEST_DIRECT"] = mode
    t = dc.RecordTable.from_reference_table(d, S, 11, arrival=False)
    ok = bool(torch.equal(t.R, ref_R) and torch.equal(t.act, ref_act))
    del t
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        t = dc.RecordTable.from_reference_table(d, S, 11, arrival=False)
        e1.record()
        est.trace(t)
        e2.record()
        torch.cuda.synchronize()
        ts.append((e0.elapsed_time(e1), e1.elapsed_time(e2)))
        del t
    print("direct" if mode == "1" else "sort  ", "equal" if ok else "MISMATCH", " ".join(f"ingest {a:.2f} + trace {b:.2f} = {a + b:.2f} ms" for a, b in ts), flush=True)
