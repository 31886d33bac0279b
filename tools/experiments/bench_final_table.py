"""The final table of an online record table: final_table_kernel (statistics stage + one evaluation per bucket) against the online
kernel without its per-record outputs, same box, same table:
    gpurun -- 'python tools/experiments/bench_final_table.py [states] [records_per_state] [actions]'"""
import os, sys, torch
sys.path.insert(0, '.')
import dcarl_amd as dc
dc.require_gpu()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
A = int(sys.argv[3]) if len(sys.argv) > 3 else 11
q = dc.workloads.sim1_q_row() if A == 11 else dc.workloads.uniform_q(1, A, seed=0)[0]
tbl = dc.sampler.sample_state_records(q, T, seed=0, stream_id=0, S=S)
est = dc.ConfidenceEstimator()
ref = None
for mode in ("1", "0", "1", "0"):
    os.environ["DCARL_FINAL_TABLE"] = mode
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b = est.bounds_from_table(tbl)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    if ref is None:
        ref = b
    same = bool(torch.equal(b.V, ref.V) and torch.equal(b.n, ref.n) and torch.equal(b.amax, ref.amax) and torch.equal(b.vmax, ref.vmax))
    ms = min(ts)
    print(f"{dc._lib.last_kernel():44s} {'same' if same else 'MISMATCH'}  {ms:.3f} ms  {5 * S * T / ms / 1e6:.0f} GB/s of the 5 B/record "
          f"({5 * S * T / ms / 1e6 / 8000:.1%} of 8 TB/s)", flush=True)
