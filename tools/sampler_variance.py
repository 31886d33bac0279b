#!/usr/bin/env python3
"""sample_pairs_kernel at 2^30 pairs read 2.17, 2.46 and 3.28 ms on three boxes of the pool — a 1.5x spread on a
store-only streaming kernel.  This script records, on ONE box, what is needed to tell "the pool" from "the kernel":

* the kernel itself (12.9 GB of stores: idx, act, R), 40 launches, per-launch HIP-event times (min / median / max);
* CONTROLS of the same size written the same way by code that is not ours: `tensor.fill_` (PyTorch's vectorised fill kernel,
  12.9 GB of plain stores) and a device-to-device copy (12.9 GB read + 12.9 GB written);
* the clocks rocm-smi reports while the kernel loops (sclk / mclk / fclk, power, temperature, performance level);
* the same kernel at 2^28 pairs (3.2 GB: a quarter of the footprint);
* both the kernel and the fill control from a COLD start: the first 64 launches after 3 s of idle, one event pair each — what a
  timing window of "2 warm-ups + 10 steps" in a fresh process actually reads.

Run it in several gpurun calls (each call is a fresh box) and compare the ratios kernel / control between boxes:
    gpurun -- 'python tools/sampler_variance.py > gpurun_out/sampler_variance_<k>.txt'"""
import json
import os
import statistics
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dcarl_amd as dc  # noqa: E402


def per_launch(fn, n=40, warm=8):
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return dict(min=t[0], median=statistics.median(t), p90=t[int(0.9 * n)], max=t[-1])


def cold_series(fn, n=64, idle_s=3.0):
    """Per-launch times of the FIRST n launches after the GPU sat idle for idle_s seconds (no warm-up): the ramp a short timing
    window reads instead of the kernel."""
    torch.cuda.synchronize()
    time.sleep(idle_s)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = [round(a.elapsed_time(b), 3) for a, b in ev]
    return dict(first_12=t[:12], mean_of_launches_1_to_12=sum(t[:12]) / 12, mean_of_launches_3_to_12=sum(t[2:12]) / 10,
                mean_of_launches_13_to_42=sum(t[12:42]) / 30, last_16_mean=sum(t[-16:]) / 16)


def smi(args):
    try:
        return subprocess.run(["rocm-smi"] + args, capture_output=True, text=True, timeout=30).stdout
    except Exception as e:   # noqa: BLE001
        return f"rocm-smi failed: {e!r}"


def main():
    dc.require_gpu()
    out = dict(device=torch.cuda.get_device_name(0), host=os.uname().nodename)
    q = dc.workloads.uniform_q(20, 11, seed=0)
    N = 1 << 30
    bufs = dc.sampler.sample_pairs(q, N, seed=0)
    out["idle_clocks"] = smi(["-c", "-P", "-t", "-p"])
    samples = []
    stop = threading.Event()

    def poll():
        while not stop.is_set():
            samples.append(smi(["-c", "-P", "--json"]))
            time.sleep(0.5)
    th = threading.Thread(target=poll, daemon=True)
    th.start()
    t_end = time.time() + 4.0
    while time.time() < t_end:                                # ~4 s of back-to-back launches under the poller
        for _ in range(50):
            dc.sampler.sample_pairs(q, N, seed=0, out=bufs)
        torch.cuda.synchronize()
    stop.set()
    th.join()
    out["clocks_under_load"] = samples[-3:]
    out["sample_pairs_2^30_ms"] = per_launch(lambda: dc.sampler.sample_pairs(q, N, seed=0, out=bufs))
    out["sample_pairs_2^30_cold_ms"] = cold_series(lambda: dc.sampler.sample_pairs(q, N, seed=0, out=bufs))
    flat = torch.empty(3 * N, dtype=torch.float32, device="cuda")                     # 12.9 GB, like the kernel's three arrays together
    out["control_fill_12.9GB_ms"] = per_launch(lambda: flat.fill_(1.5))
    out["control_fill_12.9GB_cold_ms"] = cold_series(lambda: flat.fill_(1.5))
    del flat
    a = torch.empty(3 * N, dtype=torch.float32, device="cuda")
    b = torch.empty_like(a)
    out["control_copy_12.9GB_read+write_ms"] = per_launch(lambda: b.copy_(a), n=20, warm=4)
    del a, b
    N4 = 1 << 28
    small = tuple(t[:N4] for t in bufs)
    out["sample_pairs_2^28_ms"] = per_launch(lambda: dc.sampler.sample_pairs(q, N4, seed=0, out=small))
    flat = torch.empty(3 * N4, dtype=torch.float32, device="cuda")
    out["control_fill_3.2GB_ms"] = per_launch(lambda: flat.fill_(1.5))
    k, c = out["sample_pairs_2^30_ms"]["median"], out["control_fill_12.9GB_ms"]["median"]
    out["kernel_over_fill_control"] = k / c
    out["kernel_gbs"] = 12 * N / (k * 1e-3) / 1e9
    out["fill_gbs"] = 12 * N / (c * 1e-3) / 1e9
    out["copy_gbs_read+write"] = 2 * 12 * N / (out["control_copy_12.9GB_read+write_ms"]["median"] * 1e-3) / 1e9
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
