#!/usr/bin/env python3
"""group_records (online layout -> the (state, action) bucket layout, S1:80): the write-combining kernel of round 5 against the
element-wise scatter it replaces (DCARL_GROUP_RECORDS=scatter: A/B variant of the library), same box, results compared bit for bit.
    gpurun -- 'python tools/bench_regroup.py'"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DCARL_LIB_VARIANT", "ab")
import torch  # noqa: E402

import dcarl_amd as dc  # noqa: E402


def timeit(fn, n=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def case(name, tbl):
    lib, P = dc._lib.load(), dc._lib.ptr
    n = tbl.bucket_counts()
    seg = torch.zeros(tbl.S * tbl.A + 1, dtype=torch.int64, device=tbl.device)
    torch.cumsum(n.view(-1), 0, out=seg[1:])
    total = int(seg[-1])
    fn = lib.dcarl_group_records_f32 if tbl.R.dtype == torch.float32 else lib.dcarl_group_records_f64
    outs = {}
    ms = {}
    for kind in ("sort", "4waves", "wc", "scatter"):
        if kind != "sort":
            os.environ["DCARL_GROUP_RECORDS"] = kind
        else:
            os.environ.pop("DCARL_GROUP_RECORDS", None)
        v = torch.full((max(total, 4),), float("nan"), dtype=tbl.R.dtype, device=tbl.device)

        def go():
            dc._lib.check(fn(P(tbl.R), P(tbl.act), P(tbl.slice_row_off), P(tbl.lengths), P(tbl.slot_state_i32), tbl.S, tbl.A, P(seg), P(v),
                             dc._lib.stream_ptr()), "group_records")
        ms[kind] = timeit(go, 2 if kind == "scatter" else 5)
        outs[kind] = v
    t_count = timeit(lambda: tbl.bucket_counts(), 5)
    same = all(bool(torch.equal(outs[k][:total], outs["scatter"][:total])) for k in outs) and not bool(torch.isnan(outs["sort"][:total]).any())
    es = tbl.R.element_size()
    gbs = (tbl.n_records * (2 * es + 1)) / (ms["sort"] * 1e-3) / 1e9
    print(f"{name:34s} records {tbl.n_records:>12d}  count {t_count:7.3f} ms  regroup {ms['sort']:8.3f} ms ({gbs:6.0f} GB/s)  " + "  ".join(f"[{k}] {v:.3f}" for k, v in ms.items()) + f"  "
          f"equal {same}", flush=True)


def main():
    dc.require_gpu()
    q = dc.workloads.sim1_q_row()
    case("configs[1] 65536 x 20000, A=11", dc.sampler.sample_state_records(q, 20000, seed=0, stream_id=0, S=65536))
    torch.cuda.empty_cache()
    t, _ = dc.workloads.sim2_table(2 ** 20, torch.arange(2 ** 20), A=11, mean=1000.0, seed=0)
    case("configs[3] 2^20 ragged, A=11", t)
    del t
    torch.cuda.empty_cache()
    t, _, _ = dc.workloads.mixed_records(2 ** 19, n=64, seed=0)
    case("configs[4] shard 2^19 x 16 cand.", t)
    del t
    torch.cuda.empty_cache()
    t = dc.sampler.sample_state_records(dc.workloads.uniform_q(4096, 30, seed=1), 4000, seed=1, stream_id=0, S=4096)
    case("4096 x 4000, A=30", t)
    t64 = dc.RecordTable(S=t.S, A=t.A, R=t.R.double(), act=t.act, lengths=t.lengths, slice_row_off=t.slice_row_off, n_records=t.n_records)
    case("4096 x 4000, A=30, f64", t64)


if __name__ == "__main__":
    main()
