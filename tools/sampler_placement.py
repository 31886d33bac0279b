#!/usr/bin/env python3
"""sample_pairs_kernel at 2^30 pairs on TWELVE different placements of its three output arrays inside one process (the arrays are
freed and re-allocated with other allocations of changing size in between), next to a single-stream fill of the same 12.9 GB:
what separates "the kernel's store pattern is sensitive to something" from "the pool" (VERDICT r4 item 4).
    gpurun -- 'python tools/sampler_placement.py'"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dcarl_amd as dc  # noqa: E402


def med(fn, warm=14, n=10):
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) for a, b in ev)


def main():
    dc.require_gpu()
    q = dc.workloads.uniform_q(20, 11, seed=0)
    N = 1 << 30
    rows, junk = [], []
    for k in range(12):
        bufs = dc.sampler.sample_pairs(q, N, seed=0)             # three separate allocations, like every caller
        ts = med(lambda: dc.sampler.sample_pairs(q, N, seed=0, out=bufs))
        tf = med(lambda: [b.fill_(0) for b in bufs], 4, 6)       # the same memory, one stream at a time
        rows.append((round(ts, 3), round(tf, 3), [hex(b.data_ptr()) for b in bufs]))
        print(k, rows[-1], flush=True)
        junk.append(torch.empty((k * 1237 + 400) << 20, dtype=torch.uint8, device="cuda"))
        del bufs
        torch.cuda.empty_cache()
    s = sorted(r[0] for r in rows)
    f = sorted(r[1] for r in rows)
    print("sample_pairs ms, sorted:", s, "max/min", round(s[-1] / s[0], 3))
    print("fill of the same three arrays ms, sorted:", f, "max/min", round(f[-1] / f[0], 3))


if __name__ == "__main__":
    main()
