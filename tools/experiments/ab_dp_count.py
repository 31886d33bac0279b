"""A/B of the direct ingest's count pass on one box: DCARL_DP_COUNT=queue (an item per (bucket, group), persistent blocks) against
=wide (a block per (group, 64 buckets), lane = bucket), whole ingest time and the regrouped table compared bit for bit.
    gpurun -- 'python tools/experiments/ab_dp_count.py'"""
import os, sys, torch
sys.path.insert(0, '.')
import dcarl_amd as dc
dc.require_gpu()
q = dc.workloads.sim1_q_row()


def ingest_ms(d, S, n=5):
    for _ in range(2):
        t = dc.RecordTable.from_reference_table(d, S, 11, arrival=False); del t
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        t = dc.RecordTable.from_reference_table(d, S, 11, arrival=False); del t
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


shapes = [(65536, 20000, True), (65536, 20000, False), (65536, 2048, True), (16384, 8192, True), (8192, 8192, True), (2048, 32768, True)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
for S, T, dense in shapes:
    tbl = dc.sampler.sample_state_records(q, T, seed=0, stream_id=0, S=S)
    d = tbl.to_reference_table(dense_order=True)
    ref_R, ref_act = tbl.R.clone(), tbl.act.clone()
    del tbl
    if not dense:                                  # the same rows in a uniformly random order; reference = the sort path's table
        d = d[torch.randperm(d.shape[0], generator=torch.Generator(device=d.device).manual_seed(1), device=d.device)]
        os.environ["DCARL_INGEST_DIRECT"] = "0"
        t = dc.RecordTable.from_reference_table(d, S, 11, arrival=False)
        ref_R, ref_act = t.R.clone(), t.act.clone()
        del t
    os.environ["DCARL_INGEST_DIRECT"] = "1"
    row = []
    for mode in ("queue", "wide", "queue", "wide"):
        os.environ["DCARL_DP_COUNT"] = mode
        t = dc.RecordTable.from_reference_table(d, S, 11, arrival=False)
        ok = bool(torch.equal(t.R, ref_R) and torch.equal(t.act, ref_act))
        del t
        row.append(f"{mode} {'ok' if ok else 'MISMATCH'} {ingest_ms(d, S):.2f} ms")
    print(f"S={S} T={T} {'dense' if dense else 'random'} order: " + "   ".join(row), flush=True)
    del d, ref_R, ref_act
    torch.cuda.empty_cache()
