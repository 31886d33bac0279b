"""Where do the two waves of the producer/consumer trace kernel spend their time?  Needs a library built with
-DDCARL_PAIR_PROFILE (tools/build_variant.sh) selected through DCARL_HIP_LIB, and DCARL_TRACE_KERNEL=pair."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import dcarl_amd as dc

S = int(os.environ.get("S", 65536)); T = int(os.environ.get("T", 20000))
tbl = bench.build_trace_workload(dc, S, T, 0)
est = dc.ConfidenceEstimator()
out = est.trace(tbl)
torch.cuda.synchronize()
out = est.trace(tbl, out=out)
torch.cuda.synchronize()
V = out.V.cpu().numpy().reshape(S, -1)[::64]          # first state of every slice
for name, c in (("producer", 0), ("consumer", 4)):
    tot, lg, bar, hw = V[:, c], V[:, c + 1], V[:, c + 2], V[:, c + 3].astype(np.int64)
    print(f"{name}: cycles/quad total {np.mean(tot) / (T / 4):.0f}  lgkm-drain {np.mean(lg) / (T / 4):.0f}  "
          f"barrier {np.mean(bar) / (T / 4):.0f}   (min/max total {tot.min() / (T / 4):.0f}/{tot.max() / (T / 4):.0f})")
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; se = (hw >> 13) & 7
    print("   simd histogram", np.bincount(simd, minlength=4))
hp = V[:, 3].astype(np.int64); hc = V[:, 7].astype(np.int64)
same = ((hp >> 4) & 3) == ((hc >> 4) & 3)
print("producer and consumer of a slice on the same SIMD:", same.mean())
