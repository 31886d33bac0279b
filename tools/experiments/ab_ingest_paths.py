"""Same-box A/B of the two ingest implementations (DCARL_INGEST_DIRECT=0 / 1) over table shapes: few states with long streams,
many states, skewed state popularity, small tables.   gpurun -- 'python tools/experiments/ab_ingest_paths.py'"""
import os, sys, time, torch
sys.path.insert(0, '.')
import dcarl_amd as dc
dc.require_gpu()
cases = [(1 << 20, 20, "uniform"), (1 << 22, 300, "uniform"), (1 << 26, 20, "uniform"), (1 << 26, 256, "uniform"), (1 << 26, 1000, "uniform"),
         (1 << 26, 4096, "uniform"), (1 << 26, 65536, "uniform"), (1 << 26, 65536, "skewed"), (1 << 26, 65536, "sorted"), (1 << 28, 65536, "uniform"), (1310720000, 65536, "uniform"), (1310720000, 65536, "skewed")]
if len(sys.argv) > 3:
    cases = [(int(a), int(b), c) for a, b, c in zip(sys.argv[1::3], sys.argv[2::3], sys.argv[3::3])]
for N, S, kind in cases:
    g = torch.Generator(device='cuda').manual_seed(0)
    d = torch.empty((N, 4), dtype=torch.float64, device='cuda')
    if kind == "uniform":
        st = torch.randint(0, S, (N,), generator=g, device='cuda')
    elif kind == "skewed":                       # exponential popularity: a few heavy states
        st = (torch.empty(N, device='cuda').exponential_(12.0 / S, generator=g)).long().clamp_(max=S - 1)
    elif kind.startswith("runs"):                # every state arrives in runs of L consecutive records (episodes that dwell in a state)
        L = int(kind[4:])
        st = torch.randint(0, S, ((N + L - 1) // L,), generator=g, device='cuda').repeat_interleave(L)[:N]
    else:                                        # state-major arrival: ONE bucket receives whole tiles
        st = torch.sort(torch.randint(0, S, (N,), generator=g, device='cuda')).values
    d[:, 0] = st.double()
    d[:, 1] = 0.5
    d[:, 2] = torch.randint(0, 11, (N,), generator=g, device='cuda').double()
    d[:, 3] = torch.randn(N, generator=g, device='cuda', dtype=torch.float64) * 50
    del st
    res = {}
    ref = None
    for rep in range(3):
        for mode in ("0", "1"):
            os.environ["DCARL_INGEST_DIRECT"] = mode
            torch.cuda.synchronize(); t0 = time.time()
            tbl = dc.RecordTable.from_reference_table(d, S, 11, arrival=False)
            torch.cuda.synchronize(); dt = time.time() - t0
            res.setdefault(mode, []).append(dt * 1e3)
            if ref is None: ref = (tbl.R.clone(), tbl.act.clone())
            else: assert torch.equal(tbl.R, ref[0]) and torch.equal(tbl.act, ref[1]), (N, S, kind, mode)
            del tbl
    print(f"N={N:.3g} S={S:6d} {kind:8s} sort {min(res['0']):8.2f} ms   direct {min(res['1']):8.2f} ms   ratio {min(res['0']) / min(res['1']):.2f}", flush=True)
    del d, ref
