"""How far the device's f32 Box-Muller normals are from float64 libm on the SAME Philox words — characterised, not only bounded (VERDICT r5,
weak 3).  SURVEY 8(c) asks distribution parity for the library's own RNG path (the injected-noise path is bit-exact); the other sampler tests
bound |dR|; this one records the whole error law over 2^24 draws — quantiles, and the error against |z| (the tail = small u1, where
v_log_f32 on a 24-bit u is coarsest) — into gpurun_out/sampler_normal_error.txt (copied to profiles/r06_sampler_normal_error.txt)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import c_oracle as co          # noqa: E402  (checker only)

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_box_muller_error_law_against_float64_libm():
    import dcarl_amd as dc
    dc.require_gpu()
    S, T = 4096, 4096
    tbl = dc.sampler.sample_state_records(torch.zeros((1, 1)), T, seed=2026, sigma=1.0, stream_id=9, S=S)       # Q = 0, sigma = 1: R is z itself
    z = tbl.R[tbl.state_major_index()].cpu().numpy().astype(np.float64).reshape(S, T)
    _, zr = co.sample_state_records(np.zeros((S, 1)), T, seed=2026, stream=9, sigma=1.0)
    d = np.abs(z - zr).ravel()
    za = np.abs(zr).ravel()
    lines = [f"# tests/test_sampler_normals.py: {d.size} standard normals, device f32 Box-Muller (v_log_f32 / v_cos_f32 / v_sin_f32) against float64 "
             "libm on the same Philox words",
             "|dz| quantiles: " + "  ".join(f"p{p}={np.quantile(d, p / 100):.2e}" for p in (50, 90, 99, 99.9, 99.99, 99.999)) + f"  max={d.max():.2e}",
             f"mean {z.mean():+.2e} (float64: {zr.mean():+.2e})  std {z.std():.6f} (float64: {zr.std():.6f})",
             "by |z| of the float64 draw (the tail = small u1):"]
    for lo, hi in ((0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 9)):
        m = (za >= lo) & (za < hi)
        if m.any():
            lines.append(f"  |z| in [{lo},{hi}): n={int(m.sum()):9d}  median |dz| {np.median(d[m]):.2e}  max |dz| {d[m].max():.2e}  "
                         f"max |dz|/|z| {np.max(d[m] / np.maximum(za[m], 1e-9)):.2e}")
    out = "\n".join(lines)
    print("\n" + out)
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "sampler_normal_error.txt"), "w") as f:
        f.write(out + "\n")
    assert d.max() <= 4e-6 and np.quantile(d, 0.999) <= 2e-6 and np.quantile(d, 0.5) <= 2e-7          # measured: 1.3e-6, 5.0e-7, 5.6e-8
    assert abs(z.mean() - zr.mean()) < 1e-6 and abs(z.std() - zr.std()) < 1e-6
    far = za >= 3.0                                            # the 0.27 % beyond 3 sigma: the relative error there stays tiny too
    assert np.max(d[far] / za[far]) <= 2e-5
