cd "${GRAFT_REPO_ROOT:-/root/repo}"
cat > /tmp/rg.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import dcarl_amd as dc
S = int(sys.argv[1]) if len(sys.argv) > 1 else 49152
t = dc.sampler.sample_state_records(dc.workloads.sim1_q_row(), 20000, seed=0, stream_id=0, S=S)
n = t.bucket_counts()
seg = torch.zeros(t.S * t.A + 1, dtype=torch.int64, device=t.device)
torch.cumsum(n.view(-1), 0, out=seg[1:])
v = torch.empty(int(seg[-1]), dtype=torch.float32, device=t.device)
lib, P = dc._lib.load(), dc._lib.ptr
def go():
    dc._lib.check(lib.dcarl_group_records_f32(P(t.R), P(t.act), P(t.slice_row_off), P(t.lengths), P(t.slot_state_i32), t.S, t.A, P(seg), P(v), dc._lib.stream_ptr()))
for _ in range(3): go()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): go()
e1.record(); torch.cuda.synchronize()
print(os.environ.get("DCARL_HIP_LIB", "product")[-12:], "S", S, "regroup ms", round(e0.elapsed_time(e1) / 5, 3))
PY
for lib in "" tools/ab/libRG1.so tools/ab/libRG2.so; do
  for S in 49152 16384; do
    if [ -z "$lib" ]; then python /tmp/rg.py $S 2>/dev/null; else DCARL_HIP_LIB=$PWD/$lib python /tmp/rg.py $S 2>/dev/null; fi
  done
done
