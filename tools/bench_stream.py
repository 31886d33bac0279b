"""The PCIe-inclusive rate of the online loop: a HOST-resident (N,4) f64 record table streamed through dcarl_amd.stream.trace_stream.
    gpurun -- 'python tools/bench_stream.py [states] [records_per_state] [chunk_records]'
Prints, on one box: the link's own rate (one pinned H2D copy of the table), the device-resident pass (ingest + online kernel on
the same rows already in HBM), and the streamed pipeline (registered in place / staged / with the per-record traces brought
back), each checked against the device-resident result bit for bit."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
import dcarl_amd as dc                       # noqa: E402
from dcarl_amd.stream import trace_stream    # noqa: E402

dc.require_gpu()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
CH = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 24
A = 11
q = dc.workloads.sim1_q_row()
tbl = dc.sampler.sample_state_records(q, T, seed=0, stream_id=0, S=S)
d = tbl.to_reference_table(dense_order=True)
N = d.shape[0]
del tbl
est = dc.ConfidenceEstimator()

# device-resident: the same rows already in HBM
ts = []
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ref = est.trace(dc.RecordTable.from_reference_table(d, S, A, arrival=False), want_steps=False).check()
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
dev_s = min(ts)
host = d.cpu().numpy()                       # the reference's np.load result: pageable host memory
del d
torch.cuda.empty_cache()

# the link: one copy of the whole table out of page-locked memory
pinned = torch.empty((N, 4), dtype=torch.float64, pin_memory=True)
pinned.numpy()[:] = host
dst = torch.empty((N, 4), dtype=torch.float64, device="cuda")
ts = []
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dst.copy_(pinned, non_blocking=True)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
link_s = min(ts)
# ... and what torch does with the pageable array (the one-piece route of RecordTable.from_reference_table on a host table)
t0 = time.perf_counter()
dst.copy_(torch.from_numpy(host))
torch.cuda.synchronize()
pageable_s = time.perf_counter() - t0
del dst, pinned
torch.cuda.empty_cache()

out = dict(states=S, records=N, table_bytes=N * 32, chunk_records=CH,
           device_resident_ms=dev_s * 1e3, link_copy_ms=link_s * 1e3, link_gbs=N * 32 / link_s / 1e9,
           pageable_copy_ms=pageable_s * 1e3, pageable_gbs=N * 32 / pageable_s / 1e9, runs=[])
mine = host.copy()          # pin="register" gets a private copy no pageable torch copy has touched (dcarl_amd/stream.py)
for name, kw in (("staged, 8 threads", dict()), ("staged, 8 threads", dict()), ("staged, 1 thread", dict(copy_threads=1)), ("staged, 16 threads", dict(copy_threads=16)),
                 ("registered", dict(pin="register")), ("registered", dict(pin="register")),
                 ("staged+steps_back", dict(want_steps=True))):
    r = trace_stream(mine if kw.get("pin") == "register" else host, S, A, chunk_records=CH, est=est, **kw)
    same = bool(torch.equal(r.state.V, ref.V) and torch.equal(r.state.n, ref.n) and torch.equal(r.state.act_step, ref.activation_step))
    prep = sum(t[1] for t in r.timeline)
    out["runs"].append(dict(mode=name, seconds=r.seconds, gbs=r.bytes_per_second / 1e9, records_per_s=N / r.seconds, chunks=r.chunks,
                            equals_device_resident=same, of_link_rate=link_s / r.seconds, host_prepare_s=prep))
    print(name, f"{r.seconds * 1e3:.1f} ms  {r.bytes_per_second / 1e9:.1f} GB/s  {N / r.seconds:.3e} records/s  same={same}", flush=True)
print(json.dumps(out))
