"""Why is the configs[4] shard slower per record online than equal streams?  (scratch experiment)"""
import sys, torch
sys.path.insert(0, '.')
import dcarl_amd as dc
from dcarl_amd import sampler, workloads
dc.require_gpu()
est = dc.ConfidenceEstimator()
S = 524288

def t(tbl, label):
    out = est.trace(tbl)
    est.trace(tbl, out=out); est.trace(tbl, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        est.trace(tbl, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(label, dc._lib.last_kernel(), tbl.n_records, round(ms, 3), 'ms', round(ms * 1e9 / tbl.n_records, 3), 'ps/record', flush=True)

tbl, Q, n_live = workloads.mixed_records(S, n=64, seed=0)
t(tbl, 'cfg4 as benched')
lengths = n_live.to(torch.int64) * 64
t(sampler.sample_ragged_records(Q, lengths, seed=0, n_live=n_live, sort_by_length=False), 'cfg4 unsorted slots')
t(sampler.sample_ragged_records(Q, lengths, seed=0, n_live=None), 'cfg4 lengths, all 16 candidates live')
t(sampler.sample_ragged_records(Q[1:2], lengths, seed=0, n_live=n_live), 'cfg4 lengths, one shared Q row')
l2 = torch.full((S,), 864, dtype=torch.int64)
t(sampler.sample_ragged_records(Q, l2, seed=0, n_live=n_live), 'equal 864, n_live mix')
t(sampler.sample_ragged_records(Q, l2, seed=0), 'equal 864, 16 live')
