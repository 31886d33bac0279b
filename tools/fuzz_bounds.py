"""Randomised cross-check of the final-state kernel instances against the C oracle (run on the GPU box):
    python tools/fuzz_bounds.py [iterations] [seed]
Each iteration draws S, A, a bucket-length law (Poisson / geometric / mostly empty / a few huge / tiny), a storage type,
a base-pointer offset (the values array may start at any 16-byte boundary), one compiled instance (DCARL_QUAD) or the
launcher's own choice with a random n_mean_hint, CSR or dense, and compares V, n, arg-max and max with oracle/dcarl_oracle.c."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the DCARL_* overrides this fuzzer draws exist in the A/B variant of the library only (dcarl_amd/build.py: the product .so reads no environment)
os.environ.setdefault("DCARL_LIB_VARIANT", "ab")
import numpy as np
import torch

import dcarl_amd as dc
from oracle import c_oracle as co

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
est = dc.ConfidenceEstimator()
dev = dc.require_gpu()
bad = 0
for it in range(iters):
    S = int(rng.choice([1, 2, 15, 16, 17, 31, 33, 100, 257, 1000, 4097]))
    A = int(rng.choice([1, 2, 5, 11, 12, 13, 16, 17, 30, 32]))
    law = rng.choice(["poisson", "geometric", "sparse", "spiky", "tiny", "dense"])
    nmean = int(rng.choice([1, 3, 12, 64, 91, 300, 1818]))
    B = S * A
    if law == "poisson":
        n = rng.poisson(nmean, B)
    elif law == "geometric":
        n = rng.geometric(1.0 / (nmean + 1), B) - 1
    elif law == "sparse":
        n = np.where(rng.rand(B) < 0.85, 0, rng.poisson(nmean, B))
    elif law == "spiky":
        n = rng.poisson(min(nmean, 20), B)
        n[rng.randint(0, B, 3)] = rng.randint(3000, 9000, 3)
    elif law == "tiny":
        n = rng.randint(0, 5, B)
    else:
        n = np.full(B, max(1, min(nmean, 300)))
    storage = rng.choice(["f32", "f64"])
    npdt = np.float32 if storage == "f32" else np.float64
    seg = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
    N = int(seg[-1])
    q = rng.uniform(-50, 100, B)
    sig = np.where(rng.rand(B) < 0.15, 0.0, 50.0)
    vals = (np.repeat(q, n) + np.repeat(sig, n) * rng.standard_normal(N)).astype(npdt)
    variant = rng.choice(["default", "4,4,2", "8,4,2", "4,4,1", "4,6,2", "4,6,3", "16,4,2", "0,0,0"])
    if variant == "default":
        os.environ.pop("DCARL_QUAD", None)
    else:
        os.environ["DCARL_QUAD"] = variant
    shift = int(rng.choice([0, 1, 2, 3])) * (16 // vals.itemsize)            # base pointer at another 16-byte boundary
    buf = torch.zeros(shift + max(4, N) + 8, dtype=torch.float32 if storage == "f32" else torch.float64, device=dev)
    buf[shift:shift + N] = torch.from_numpy(vals).to(dev)
    view = buf[shift:]
    hint = int(rng.choice([0, 1, 16, 64, 256, 5000]))
    if law == "dense":
        res = est.bounds(view, S, A, n_dense=int(n[0]), n_mean_hint=hint)
    else:
        res = est.bounds(view, S, A, seg_off=torch.from_numpy(seg), n_mean_hint=hint)
    ref = co.bounds_csr(vals if N else np.zeros(4, npdt), seg, S, A)
    V, Vr = res.V.cpu().numpy(), ref["V"]
    okV = np.allclose(V, Vr, rtol=1e-10, atol=1e-10)
    checks = dict(V=okV, n=np.array_equal(res.n.cpu().numpy(), ref["n"]), amax=np.array_equal(res.amax.cpu().numpy(), ref["amax"]),
                  vmax=np.allclose(res.vmax.double().cpu().numpy(), ref["vmax"].astype(np.float64), rtol=1e-6, atol=1e-6))
    ok = all(checks.values())
    bad += not ok
    if not ok:
        print({k: v for k, v in checks.items() if not v}, "worst |dV|", float(np.abs(V - Vr).max()))
    print(f"{it:3d} S={S:5d} A={A:2d} {law:9s} n~{nmean:4d} {storage} {variant:7s} hint={hint:4d} shift={shift} N={N:8d} "
          f"{dc._lib.last_kernel()} {'ok' if ok else 'MISMATCH'}", flush=True)
print("all ok" if not bad else f"{bad} MISMATCHES")
