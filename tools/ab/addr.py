import os, sys, torch, statistics
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import dcarl_amd as dc
q = dc.workloads.uniform_q(20, 11, seed=0)
N = 1 << 30
def t(bufs, n=12):
    for _ in range(16): dc.sampler.sample_pairs(q, N, seed=0, out=bufs)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); dc.sampler.sample_pairs(q, N, seed=0, out=bufs); b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) for a, b in ev)
big = torch.empty(3 * N * 4 + (64 << 20), dtype=torch.uint8, device="cuda")
base = big.data_ptr()
print("base % 2MB", base % (2 << 20), "base % 1GB", base % (1 << 30))
for skew in (0, 256, 1024, 4096, 16384, 65536, 1 << 20, 3 << 20, (1 << 20) + 4096 + 256):
    o = [0, 4 * N + skew, 8 * N + 2 * skew]
    idx = big[o[0]:o[0] + 4 * N].view(torch.int32); act = big[o[1]:o[1] + 4 * N].view(torch.int32); R = big[o[2]:o[2] + 4 * N].view(torch.float32)
    print("skew", skew, round(t((idx, act, R)), 3), flush=True)
del big
torch.cuda.empty_cache()
# separately allocated (what the bench does), a few times with other allocations in between
junk = []
for k in range(6):
    bufs = dc.sampler.sample_pairs(q, N, seed=0)
    print("separate alloc", k, [hex(b.data_ptr()) for b in bufs], round(t(bufs), 3), flush=True)
    junk.append(torch.empty((k + 1) * 977 * (1 << 20), dtype=torch.uint8, device="cuda"))
    del bufs
