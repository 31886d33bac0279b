"""One ingest of the configs[1] table per path, for rocprofv3 --kernel-trace --stats (tools: per-kernel times of the chain)."""
import os, sys, torch
sys.path.insert(0, '.')
import dcarl_amd as dc
dc.require_gpu()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
mode = sys.argv[2] if len(sys.argv) > 2 else "1"
q = dc.workloads.sim1_q_row()
tbl = dc.sampler.sample_state_records(q, 20000, seed=0, stream_id=0, S=S)
d = tbl.to_reference_table(dense_order=True)
del tbl
os.environ["DCARL_INGEST_DIRECT"] = mode
for _ in range(3):
    t = dc.RecordTable.from_reference_table(d, S, 11, arrival=False)
    del t
torch.cuda.synchronize()
