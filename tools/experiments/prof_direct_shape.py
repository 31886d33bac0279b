"""One direct-path ingest of a synthetic (N, S, kind) table for rocprofv3 --kernel-trace --stats."""
import os, sys, torch
sys.path.insert(0, '.')
import dcarl_amd as dc
dc.require_gpu()
N, S, kind = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
g = torch.Generator(device='cuda').manual_seed(0)
d = torch.empty((N, 4), dtype=torch.float64, device='cuda')
if kind == "uniform":
    st = torch.randint(0, S, (N,), generator=g, device='cuda')
elif kind == "skewed":
    st = (torch.empty(N, device='cuda').exponential_(12.0 / S, generator=g)).long().clamp_(max=S - 1)
else:
    st = torch.sort(torch.randint(0, S, (N,), generator=g, device='cuda')).values
d[:, 0] = st.double(); d[:, 1] = 0.5
d[:, 2] = torch.randint(0, 11, (N,), generator=g, device='cuda').double()
d[:, 3] = torch.randn(N, generator=g, device='cuda', dtype=torch.float64) * 50
os.environ["DCARL_INGEST_DIRECT"] = sys.argv[4] if len(sys.argv) > 4 else "1"
for _ in range(3):
    t = dc.RecordTable.from_reference_table(d, S, 11, arrival=False); del t
torch.cuda.synchronize()
