"""final_table_kernel: 1.25 ms on a sampler-made table, 1.77 ms inside bounds_from_reference_table's chain — which part of the difference is
the table (an ingest-made table: recycled allocations, a slot map) and which the kernels that ran just before it?"""
import sys, torch
sys.path.insert(0, '.')
import dcarl_amd as dc
dc.require_gpu()
S, T, A = 65536, 20000, 11
q = dc.workloads.sim1_q_row()
src = dc.sampler.sample_state_records(q, T, seed=0, stream_id=0, S=S)
d = src.to_reference_table(dense_order=True)
est = dc.ConfidenceEstimator()


def t_ms(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print("sampler-made table, final table alone      ", round(t_ms(lambda: est.bounds_from_table(src)), 3), "ms")
ing = dc.RecordTable.from_reference_table(d, S, A, arrival=False)
print("ingest-made table (sorted slots), alone     ", round(t_ms(lambda: est.bounds_from_table(ing)), 3), "ms")
ing2 = dc.RecordTable.from_reference_table(d, S, A, arrival=False, sort_by_length=False)
print("ingest-made table (identity slots), alone   ", round(t_ms(lambda: est.bounds_from_table(ing2)), 3), "ms")
del ing, ing2
# inside the chain: events around the final-table call only
ts = []
for _ in range(8):
    t = dc.RecordTable.from_reference_table(d, S, A, arrival=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); est.bounds_from_table(t); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
    del t
print("right after the ingest of the same table    ", [round(x, 3) for x in ts])

# the online kernel on the same three tables (does the slot map cost it anything?)
ing = dc.RecordTable.from_reference_table(d, S, A, arrival=False)
ing2 = dc.RecordTable.from_reference_table(d, S, A, arrival=False, sort_by_length=False)
for name, t in (("sampler-made", src), ("ingest-made, sorted slots", ing), ("ingest-made, identity slots", ing2)):
    o = est.trace(t)
    print(f"online kernel, {name:28s}", round(t_ms(lambda: est.trace(t, out=o), n=20, warm=10), 3), "ms")
