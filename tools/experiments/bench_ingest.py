"""Record ingest: the reference's (N,4) f64 table in ARRIVAL order -> the sliced per-state layout (RecordTable.from_reference_table).
    gpurun -- 'python tools/experiments/bench_ingest.py'"""
import sys, time, torch
sys.path.insert(0, '.')
import dcarl_amd as dc
dc.require_gpu()
cases = [(1 << 24, 4096), (1 << 26, 65536), (1 << 26, 1 << 20), (1 << 26, 20)]
if len(sys.argv) > 1:
    cases = [(int(sys.argv[1]), int(sys.argv[2]))]
for N, S in cases:
    g = torch.Generator(device='cuda').manual_seed(0)
    d = torch.empty((N, 4), dtype=torch.float64, device='cuda')
    d[:, 0] = torch.randint(0, S, (N,), generator=g, device='cuda').double()
    d[:, 1] = torch.rand(N, generator=g, device='cuda', dtype=torch.float64)
    d[:, 2] = torch.randint(0, 11, (N,), generator=g, device='cuda').double()
    d[:, 3] = torch.randn(N, generator=g, device='cuda', dtype=torch.float64) * 50
    for arrival in (False, True):
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.time()
            tbl = dc.RecordTable.from_reference_table(d, S, 11, arrival=arrival)
            torch.cuda.synchronize(); dt = time.time() - t0
            print(N, S, 'arrival' if arrival else 'plain', 'from_reference_table %.2f ms = %.2e records/s = %.0f GB/s of 37 B/record' %
                  (dt * 1e3, N / dt, N * 37 / dt / 1e9), flush=True)
            del tbl
    est = dc.ConfidenceEstimator()
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        r = est.bounds_from_reference_table(d, S, 11)
        torch.cuda.synchronize(); dt = time.time() - t0
        print(N, S, 'bounds_from_reference_table %.2f ms' % (dt * 1e3), flush=True)
    del d
