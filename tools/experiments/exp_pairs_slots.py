"""sampler -> from_pairs -> online loop with and without length-sorted slots (same box):
    gpurun -- 'python tools/experiments/exp_pairs_slots.py [states] [pairs]'"""
import sys, torch
sys.path.insert(0, '.')
import dcarl_amd as dc
dc.require_gpu()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 30
A = 11
q = dc.workloads.uniform_q(S, A, seed=0)
est = dc.ConfidenceEstimator()
pairs = dc.sampler.sample_pairs(q, N, seed=0)
ref = None
for sort in (True, False, True, False):
    ts = []
    for _ in range(3):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        t = dc.RecordTable.from_pairs(*pairs, S, A, sort_by_length=sort)
        e[1].record()
        tr = est.trace(t)
        e[2].record()
        torch.cuda.synchronize()
        ts.append((e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])))
    if ref is None:
        ref = tr
    same = bool(torch.equal(tr.V, ref.V) and torch.equal(tr.activation_step, ref.activation_step) and torch.equal(tr.amax, ref.amax))
    print("sorted slots  " if sort else "identity slots", f"rows {t.rows}", "same results" if same else "MISMATCH",
          " ".join(f"ingest {a:.2f} + online {b:.2f} = {a + b:.2f} ms" for a, b in ts), flush=True)
    del t, tr
