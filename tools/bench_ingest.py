"""Record ingest: the reference's (N,4) f64 table in ARRIVAL order -> the sliced per-state layout (RecordTable.from_reference_table).
    gpurun -- 'python tools/bench_ingest.py'"""
import sys, time, torch
sys.path.insert(0, '.')
import dcarl_amd as dc
dc.require_gpu()
for N, S in ((1 << 24, 4096), (1 << 26, 65536)):
    g = torch.Generator(device='cuda').manual_seed(0)
    d = torch.empty((N, 4), dtype=torch.float64, device='cuda')
    d[:, 0] = torch.randint(0, S, (N,), generator=g, device='cuda').double()
    d[:, 1] = torch.rand(N, generator=g, device='cuda', dtype=torch.float64)
    d[:, 2] = torch.randint(0, 11, (N,), generator=g, device='cuda').double()
    d[:, 3] = torch.randn(N, generator=g, device='cuda', dtype=torch.float64) * 50
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        tbl = dc.RecordTable.from_reference_table(d, S, 11)
        torch.cuda.synchronize(); dt = time.time() - t0
        print(N, S, 'from_reference_table %.1f ms = %.2e records/s' % (dt * 1e3, N / dt), flush=True)
        del tbl
    del d
