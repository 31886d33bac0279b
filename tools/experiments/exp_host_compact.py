"""dcarl_host_compact_rows_f32 alone: rows per second against the number of host threads (no GPU involved), next to a plain threaded copy
of the same rows (what the uncompacted staging does)."""
import os, sys, time
import numpy as np
from concurrent.futures import ThreadPoolExecutor
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from dcarl_amd.records import compact_rows_host
N = int(os.environ.get("ROWS", 1 << 25))
rng = np.random.default_rng(0)
rows = np.empty((N, 4))
rows[:, 0] = rng.integers(0, 65536, N); rows[:, 1] = 0.5; rows[:, 2] = rng.integers(0, 11, N); rows[:, 3] = rng.normal(20, 50, N)
out = np.empty(N, np.int64)
dst = np.empty_like(rows)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for th in (1, 2, 4, 8, 16, 32, 64, 128):
    if th > 2 * (os.cpu_count() or 1):
        break
    with ThreadPoolExecutor(th) as pool:
        best = bestc = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); compact_rows_host(rows, 65536, 11, out=out, pool=pool, pieces=th); best = min(best, time.perf_counter() - t0)
            step = -(-N // th)
            t0 = time.perf_counter()
            for f in [pool.submit(np.copyto, dst[i:i + step], rows[i:i + step]) for i in range(0, N, step)]:
                f.result()
            bestc = min(bestc, time.perf_counter() - t0)
    print(f"threads {th:3d}: compact {N / best / 1e9:6.2f} Grows/s ({N * 32 / best / 1e9:6.1f} GB/s read)   copy {N * 32 / bestc / 1e9:6.1f} GB/s", flush=True)
