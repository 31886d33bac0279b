"""Randomised cross-check of the record ingest (csrc/ingest.hip) against a stable NumPy sort (run on the GPU box):
    python tools/fuzz_ingest.py [iterations] [seed]
Each iteration draws the table size, the number of states (1 ... 300 000) and actions, an arrival law (uniform / heavy-tailed /
long runs / few states / sorted), the storage type, slot sorting and arrival bookkeeping on or off, both scatter instances, and
checks every output of from_reference_table (lengths, slot order, row offsets, every state's records in arrival order, the
arrival bookkeeping) and of the (state, action) grouping bit for bit."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the DCARL_* overrides this fuzzer draws exist in the A/B variant of the library only (dcarl_amd/build.py: the product .so reads no environment)
os.environ.setdefault("DCARL_LIB_VARIANT", "ab")
import numpy as np
import torch

import dcarl_amd as dc
from dcarl_amd import _lib
from dcarl_amd.records import as_device_table, check_ingest_info

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
lib = dc.load_library()
dev = dc.require_gpu()
for it in range(iters):
    S = int(rng.choice([1, 2, 63, 64, 65, 200, 255, 256, 257, 4096, 5000, 65535, 65536, 65537, 300000]))
    A = int(rng.choice([1, 2, 11, 16, 17, 32]))
    N = int(rng.choice([0, 1, 7, 100, 4095, 6656, 6657, 8192, 8193, 20000, 100000, 300000, 1500000, 6000000]))
    law = rng.choice(["uniform", "zipf", "runs", "few", "sorted"])
    if law == "uniform":
        st = rng.randint(0, S, N)
    elif law == "zipf":
        st = np.minimum(rng.zipf(1.3, N) - 1, S - 1)
    elif law == "runs":
        st = np.repeat(rng.randint(0, S, N // 50 + 1), 50)[:N]
    elif law == "few":
        st = rng.choice(rng.randint(0, S, 3), N) if N else np.zeros(0, dtype=np.int64)
    else:
        st = np.sort(rng.randint(0, S, N))
    d = np.empty((N, 4))
    d[:, 0] = st + rng.rand(N) * 0.5
    d[:, 1] = rng.rand(N)
    ac = rng.randint(0, A, N)
    d[:, 2] = ac
    d[:, 3] = rng.normal(0, 50, N)
    storage = torch.float32 if rng.rand() < 0.5 else torch.float64
    npdt = np.float32 if storage == torch.float32 else np.float64
    sort_len, arrival = bool(rng.rand() < 0.7), bool(rng.rand() < 0.4)
    os.environ["DCARL_INGEST_SCATTER_THREADS"] = str(rng.choice(["256", "512"]))
    os.environ["DCARL_INGEST_PAIRS"] = str(rng.choice(["1", "1", "0"]))      # f32 without arrival: the pair-record passes, or not
    # the direct path (f32, no arrival bookkeeping, <= 65 536 states): the automatic choice, forced at every size, or never; its count
    # pass in either form or the launcher's choice
    for var, val in (("DCARL_INGEST_DIRECT", rng.choice(["", "1", "1", "0"])), ("DCARL_DP_COUNT", rng.choice(["", "queue", "wide"]))):
        if val:
            os.environ[var] = str(val)
        else:
            os.environ.pop(var, None)
    tbl = dc.RecordTable.from_reference_table(d, S, A, storage=storage, sort_by_length=sort_len, arrival=arrival)
    counts = np.bincount(st, minlength=S)
    order = np.argsort(st, kind="stable")
    ok = dict(lengths=np.array_equal(tbl.lengths_by_state.cpu().numpy(), counts))
    if sort_len and S > 64:
        ok["slots"] = np.array_equal(tbl.slot_state.cpu().numpy(), np.argsort(-counts, kind="stable"))
    idx = tbl.state_major_index()
    ok["R"] = np.array_equal(tbl.R[idx].cpu().numpy(), d[order, 3].astype(npdt))
    ok["act"] = np.array_equal(tbl.act[idx].cpu().numpy(), ac[order].astype(np.uint8))
    rows = int(tbl.slice_row_off[-1].item())
    ok["padding"] = rows == 0 or np.count_nonzero(tbl.R[:rows * 64].cpu().numpy()) == np.count_nonzero(d[:, 3].astype(npdt))
    if arrival:
        ok["rec_elem"] = np.array_equal(tbl.R[tbl.rec_elem].cpu().numpy(), d[:, 3].astype(npdt)) and np.array_equal(tbl.rec_state.cpu().numpy(), st)
        off = np.concatenate([[0], np.cumsum(counts)])
        t_ref = np.empty(N, dtype=np.int64)
        t_ref[order] = np.arange(N) - off[st[order]]
        ok["rec_t"] = np.array_equal(tbl.rec_t.cpu().numpy(), t_ref)
    # the final-state form
    f32 = storage == torch.float32
    dd = as_device_table(d, dev)
    ws = torch.empty(int(lib.dcarl_ingest_workspace_bytes(N, S, A, 4 if f32 else 8, 0, 1)), dtype=torch.uint8, device=dev)
    vals = torch.zeros(max(N, 4), dtype=storage, device=dev)
    seg = torch.empty(S * A + 1, dtype=torch.int64, device=dev)
    info = torch.empty(16, dtype=torch.int64, device=dev)
    fn = lib.dcarl_ingest_buckets_f32 if f32 else lib.dcarl_ingest_buckets_f64
    _lib.check(fn(_lib.ptr(dd), N, S, A, _lib.ptr(ws), _lib.ptr(vals), _lib.ptr(seg), _lib.ptr(info), _lib.stream_ptr()))
    check_ingest_info(info, S, A, N)
    key = st.astype(np.int64) * A + ac
    ok["seg"] = np.array_equal(seg.cpu().numpy(), np.concatenate([[0], np.cumsum(np.bincount(key, minlength=S * A))]))
    ok["values"] = np.array_equal(vals[:N].cpu().numpy(), d[np.argsort(key, kind="stable"), 3].astype(npdt))
    # round 5: the same buckets by ingest + the chunk-sort regroup (records.buckets_from_reference_table), bit for bit
    from dcarl_amd.records import buckets_from_reference_table
    rv, rs = buckets_from_reference_table(dd, S, A, storage=storage, via="regroup")
    ok["regroup_seg"] = bool(torch.equal(rs, seg))
    ok["regroup_values"] = bool(torch.equal(rv[:N], vals[:N]))
    good = all(ok.values())
    print(f"{it:3d} S={S:6d} A={A:2d} N={N:8d} {law:8s} {'f32' if f32 else 'f64'} sort={int(sort_len)} arrival={int(arrival)} "
          f"threads={os.environ['DCARL_INGEST_SCATTER_THREADS']} pairs={os.environ['DCARL_INGEST_PAIRS']} direct={os.environ.get('DCARL_INGEST_DIRECT', '-')} count={os.environ.get('DCARL_DP_COUNT', '-')} {'ok' if good else 'MISMATCH ' + str([k for k, v in ok.items() if not v])}", flush=True)
    if not good:
        sys.exit(1)
print("all ok")
