"""Soak test of the record ingest at BASELINE configs[1] size (run on the GPU box): the same 42-GB arrival-ordered table is
regrouped again and again — table mode with and without arrival bookkeeping, bucket mode — and every result is compared with the
first one bit for bit (and the first one with the source table).   python tools/soak_ingest.py [iterations] [states]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dcarl_amd as dc

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
S = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
T = 20000
dc.require_gpu()
q = dc.workloads.sim1_q_row()
src = dc.sampler.sample_state_records(q, T, seed=0, stream_id=0, S=S)
d = src.to_reference_table(dense_order=True)
est = dc.ConfidenceEstimator()
ref = dc.RecordTable.from_reference_table(d, S, 11, arrival=False)
assert torch.equal(ref.R, src.R) and torch.equal(ref.act, src.act), "first regrouping differs from the source"
refb = est.bounds_from_reference_table(d, S, 11)
refa = dc.RecordTable.from_reference_table(d, S, 11, arrival=True)
assert torch.equal(refa.R, src.R) and torch.equal(refa.R[refa.rec_elem], d[:, 3].float())
ea, ta = refa.rec_elem.clone(), refa.rec_t.clone()
del refa
bad = 0
t0 = time.time()
for it in range(iters):
    t = dc.RecordTable.from_reference_table(d, S, 11, arrival=False)
    ok = torch.equal(t.R, ref.R) and torch.equal(t.act, ref.act) and torch.equal(t.lengths, ref.lengths)
    del t
    b = est.bounds_from_reference_table(d, S, 11)
    ok = ok and torch.equal(b.V, refb.V) and torch.equal(b.n, refb.n) and torch.equal(b.amax, refb.amax)
    del b
    if it % 5 == 0:
        a = dc.RecordTable.from_reference_table(d, S, 11, arrival=True)
        ok = ok and torch.equal(a.R, ref.R) and torch.equal(a.rec_elem, ea) and torch.equal(a.rec_t, ta)
        del a
    bad += not ok
    if not ok or it % 10 == 0:
        print(it, "ok" if ok else "MISMATCH", f"{time.time() - t0:.1f} s", flush=True)
print("soak:", iters, "iterations,", bad, "mismatches")
sys.exit(1 if bad else 0)
