"""CPU-side checks: the C-ABI library loads, exports every symbol include/dcarl.h declares, the ctypes table
mirrors the header, argument validation returns error codes (no GPU needed), host layout math, no-fallback rule."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import dcarl_amd
from dcarl_amd import _lib, layout

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(REPO, "include", "dcarl.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int32_t|int64_t|void|const char\*)\s+(dcarl_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("void", "") else len(args.split(","))
    return out


def test_every_declared_symbol_is_exported_and_bound():
    decl = header_functions()
    assert len(decl) >= 20
    dcarl_amd.load_library()            # first: a stale library is rebuilt in place here, and a handle mapped BEFORE that would stay the old file's
    lib = C.CDLL(_lib.LIB_PATH)
    for name, nargs in decl.items():
        assert hasattr(lib, name), f"{name} declared in include/dcarl.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} not bound in dcarl_amd/_lib.py"
        assert len(_lib.SIGNATURES[name][1]) == nargs, name
    assert set(_lib.SIGNATURES) == set(decl)
    assert dcarl_amd.load_library().dcarl_version() == _lib.ABI_VERSION == 7
    from dcarl_amd import build
    assert dcarl_amd.load_library().dcarl_build_id().decode() == build.source_id()      # no stale library


def test_struct_layout_matches_header():
    assert C.sizeof(_lib.CParams) == 48 and _lib.CParams.alpha.offset == 8 and _lib.CParams.init_other.offset == 40
    p = _lib.CParams()
    dcarl_amd.load_library().dcarl_default_params(C.byref(p))
    assert (p.rule_act, p.n_thres, p.alpha, p.scale, p.cap, p.init_rule, p.init_other) == (0, 10, 0.05, 150.0, 100.0, 100.0, -50.0)
    assert dcarl_amd.Params().to_c().scale == 150.0


def test_argument_validation_without_gpu():
    lib = dcarl_amd.load_library()
    p = dcarl_amd.Params().to_c()
    null = C.c_void_p(None)
    one = C.c_void_p(16)
    rc = lib.dcarl_trace_f32(one, one, one, one, null, 4, 0, C.byref(p), null, null, null, null, null, null, null, null)
    assert rc == -1 and b"A=0" in lib.dcarl_last_error()
    rc = lib.dcarl_trace_f32(one, one, one, one, null, 4, 33, C.byref(p), null, null, null, null, null, null, null, null)
    assert rc == -1
    rc = lib.dcarl_trace_f32(null, one, one, one, null, 4, 11, C.byref(p), null, null, null, null, null, null, null, null)
    assert rc == -1 and b"non-NULL" in lib.dcarl_last_error()
    rc = lib.dcarl_trace_f32(C.c_void_p(4), one, one, one, null, 4, 11, C.byref(p), null, null, null, null, null, null, null, null)
    assert rc == -1 and b"alignment" in lib.dcarl_last_error()
    bad = dcarl_amd.Params(rule_act=11).to_c()
    assert lib.dcarl_trace_f64(one, one, one, one, null, 4, 11, C.byref(bad), null, null, null, null, null, null, null, null) == -1
    bad = dcarl_amd.Params(alpha=1.5).to_c()
    assert lib.dcarl_bounds_csr_f32(one, null, 4, 0, 1, 11, C.byref(bad), null, null, null, null, null) == -1
    assert lib.dcarl_trace_f32(one, one, one, one, null, 0, 11, C.byref(p), null, null, null, null, null, null, null, null) == 0
    assert lib.dcarl_trace_f32(one, one, one, one, null, 1, 11, None, null, null, null, null, null, null, null, null) == -1
    assert lib.dcarl_scan_f64(null, null, 5, null, null) == -1
    assert lib.dcarl_scan_workspace_bytes(5000) >= 3 * 8
    assert lib.dcarl_sample_pairs(one, 0, 11, 5, 50.0, 1, 0, 0, one, one, one, null, null) == -1
    # the entry points added with ABI version 2
    assert lib.dcarl_count_records(one, one, one, null, 4, 33, one, null) == -1 and b"A=33" in lib.dcarl_last_error()
    assert lib.dcarl_count_records(one, one, one, null, 4, 11, null, null) == -1
    assert lib.dcarl_group_records_f32(one, one, one, one, null, 4, 11, null, one, null) == -1
    assert lib.dcarl_group_records_f64(one, one, one, one, null, 0, 11, null, null, null) == 0
    assert lib.dcarl_sample_state_records_ragged(one, 1, 4, 11, one, 6, one, null, null, 50.0, 1, 0, 0, null, one, one, null) == -1
    assert b"multiple of 4" in lib.dcarl_last_error()
    assert lib.dcarl_sample_state_records_ragged(one, 2, 4, 11, one, 8, one, null, null, 50.0, 1, 0, 0, null, one, one, null) == -1
    assert lib.dcarl_sample_buckets(one, 1, 4, 11, null, -1, 50.0, 1, 2, one, null) == -1
    assert lib.dcarl_sample_buckets(one, 1, 4, 11, null, 8, 50.0, 1, 2, C.c_void_p(4), null) == -1
    assert lib.dcarl_workspace_bytes(1, 0, 0, 5000) == lib.dcarl_scan_workspace_bytes(5000)
    assert lib.dcarl_workspace_bytes(2, 16, 0, 1000) == lib.dcarl_rls_workspace_bytes(1000, 16)
    assert lib.dcarl_workspace_bytes(99, 1, 1, 1) == 0
    assert lib.dcarl_workspace_bytes(4, 1000, 0, 0) == 4 * 272 and lib.dcarl_workspace_bytes(4, 10 ** 7, 0, 0) == 512 * 272
    assert lib.dcarl_summary_stats(one, one, one, 5, 33, one, one, null) == -1 and b"A=33" in lib.dcarl_last_error()
    assert lib.dcarl_summary_stats(one, one, one, 5, 11, one, null, null) == -1
    assert lib.dcarl_workspace_bytes(3, 0, 0, 1000) >= 2048 * 16 + 1000 * 4 + 16 * 12   # table slots, row -> slot, bit words + their prefix
    assert lib.dcarl_state_ids(one, null, 5, 65, 0, one, one, one, null) == -1 and b"D=65" in lib.dcarl_last_error()
    assert lib.dcarl_state_ids(one, null, 5, 20, 0, one, one, null, null) == -1
    assert lib.dcarl_state_ids(one, null, 5, 20, 0, C.c_void_p(8), one, one, null) == -1 and b"alignment" in lib.dcarl_last_error()
    assert lib.dcarl_state_ids(one, null, 5, 20, -1, one, one, one, null) == -1
    # ABI version 5: cells + ids in one call
    assert lib.dcarl_index_states_f64(one, 5, 18, one, 0, one, one, one, one, null) == -1 and b"multiple of 4" in lib.dcarl_last_error()
    assert lib.dcarl_index_states_f64(one, 5, 20, one, 0, one, one, one, null, null) == -1
    assert lib.dcarl_index_states_f64(one, 5, 20, one, 0, C.c_void_p(8), one, one, one, null) == -1 and b"alignment" in lib.dcarl_last_error()
    assert lib.dcarl_index_states_f64(null, 0, 20, null, 0, null, null, null, one, null) == 0
    assert lib.dcarl_workspace_bytes(3, 100, 0, 10 ** 6) < lib.dcarl_workspace_bytes(3, 0, 0, 10 ** 6)
    assert lib.dcarl_state_cells_f64(one, 5, 3, one, one, one, null) == -1 and b"D % 4" in lib.dcarl_last_error()
    assert lib.dcarl_episode_returns_f64(one, one, one, one, -1, null, one, null, null) == -1
    assert lib.dcarl_nstep_backup_f64(one, one, one, 3, null, 10, one, null, null) == -1
    assert lib.dcarl_nstep_backup_f64(null, null, null, 0, null, 10, null, null, null) == 0
    # record ingest (ABI version 4)
    ws = C.c_void_p(256)
    assert lib.dcarl_ingest_workspace_bytes(1 << 20, 4096, 11, 4, 3, 0) >= (1 << 20) * 2 * 12
    assert lib.dcarl_ingest_workspace_bytes(1 << 20, 4096, 11, 8, 0, 0) >= (1 << 20) * 2 * 12
    assert lib.dcarl_ingest_workspace_bytes(1 << 20, 4096, 11, 5, 0, 0) == 0 and lib.dcarl_ingest_workspace_bytes(2 ** 31, 1, 1, 4, 0, 0) == 0
    assert lib.dcarl_workspace_bytes(5, 4096, 11, 1 << 20) >= lib.dcarl_ingest_workspace_bytes(1 << 20, 4096, 11, 4, 3, 0)
    assert lib.dcarl_workspace_bytes(6, 4096, 11, 1 << 20) >= lib.dcarl_ingest_workspace_bytes(1 << 20, 4096, 11, 8, 0, 1)
    assert lib.dcarl_ingest_group_f32(C.c_void_p(32), 2 ** 31, 4, 11, 0, ws, one, one, one, one, null, one, null) == -1 and b"2^31" in lib.dcarl_last_error()
    assert lib.dcarl_ingest_group_f32(C.c_void_p(32), 5, 4, 33, 0, ws, one, one, one, one, null, one, null) == -1 and b"A=33" in lib.dcarl_last_error()
    assert lib.dcarl_ingest_group_f32(C.c_void_p(32), 5, 4, 11, 0, C.c_void_p(16), one, one, one, one, null, one, null) == -1
    assert lib.dcarl_ingest_group_f64(C.c_void_p(8), 5, 4, 11, 0, ws, one, one, one, one, null, one, null) == -1 and b"32-byte" in lib.dcarl_last_error()
    assert lib.dcarl_ingest_group_f64(C.c_void_p(32), 5, 4, 11, 2, ws, one, one, one, one, null, one, null) == -1 and b"rec_state" in lib.dcarl_last_error()
    assert lib.dcarl_ingest_pack_f32(5, 4, 11, 0, ws, one, null, one, 0, one, one, null, null, null) == 0
    assert lib.dcarl_ingest_pack_f32(5, 4, 11, 0, ws, one, null, one, 99, one, one, null, null, null) == -1 and b"total_bands" in lib.dcarl_last_error()
    assert lib.dcarl_ingest_pack_f64(5, 4, 11, 2, ws, one, null, one, 1, one, one, null, null, null) == -1 and b"rec_elem" in lib.dcarl_last_error()
    assert lib.dcarl_ingest_buckets_f32(C.c_void_p(32), 5, 4, 11, ws, null, one, one, null) == -1
    assert lib.dcarl_slot_order_workspace_bytes(1000) > 4 * 1000 * 4 and lib.dcarl_slot_order_workspace_bytes(0) == 0
    assert lib.dcarl_slot_order(one, 0, 5, 1, ws, one, one, one, one, one, null) == -1
    assert lib.dcarl_slot_order(one, 5, -1, 1, ws, one, one, one, one, one, null) == -1
    assert lib.dcarl_slot_order(one, 5, 5, 1, C.c_void_p(8), one, one, one, one, one, null) == -1
    assert lib.dcarl_count_nonfinite(one, 3, 5, one, null) == -1 and lib.dcarl_count_nonfinite(null, 4, 5, one, null) == -1
    assert lib.dcarl_rls_gate_train(one, one, one, null, 5, 30, one, null, null) == -1          # an action output needs rl_action
    assert lib.dcarl_rls_gate_train(one, one, one, one, 0, 30, one, null, null) == 0
    assert lib.dcarl_export_records_f32(one, one, one, null, null, 4, 5, 3, null, null, 21, C.c_void_p(32), null) == -1     # N != S * T
    assert lib.dcarl_export_records_f32(one, one, one, null, null, 4, 5, 2, null, null, 20, C.c_void_p(32), null) == -1 and b"coprime" in lib.dcarl_last_error()
    import threading
    seen = []
    t = threading.Thread(target=lambda: seen.append(lib.dcarl_last_kernel()))      # (a thread of its own: other tests of this process launch)
    t.start(); t.join()
    assert seen == [b""]                                         # nothing launched on that thread yet
    assert lib.dcarl_allgather_summary(null, one, one, 12, null) == -1
    assert lib.dcarl_comm_init(2, 2, one, C.pointer(C.c_void_p())) == -1 and b"rank 2 of 2" in lib.dcarl_last_error()
    assert lib.dcarl_comm_destroy(null) == 0


def test_product_path_has_no_cpu_fallback():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(dcarl_amd.DcarlError):
        dcarl_amd.reference_api.upper_bound(np.arange(20.0))
    with pytest.raises(dcarl_amd.DcarlError):
        dcarl_amd.RecordTable.from_reference_table(np.zeros((3, 4)), 1, 11)
    with pytest.raises(dcarl_amd.DcarlError):
        dcarl_amd.sampler.sample_pairs(torch.zeros((20, 11)), 10, 0)
    # and the package never imports the oracle
    import subprocess, sys
    code = "import sys, dcarl_amd; assert not [m for m in sys.modules if m.startswith('oracle')]"
    subprocess.check_call([sys.executable, "-c", code], cwd=REPO)
    for root, _, files in os.walk(os.path.join(REPO, "dcarl_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_layout_math():
    lens = torch.tensor([5, 0, 9] + [1] * 70)
    off = layout.slice_row_offsets(lens)
    assert off.tolist() == [0, 12, 16]
    assert layout.slice_row_offsets(torch.zeros(0, dtype=torch.int64)).tolist() == [0]
    s = torch.tensor([0, 0, 0, 2, 2, 64, 72])
    t = torch.tensor([0, 3, 4, 8, 7, 0, 0])
    e = layout.elem_index(off, s, t)
    assert e.tolist() == [0, 3, 256, 2 * 64 * 4 + 2 * 4, 64 * 4 + 2 * 4 + 3, 12 * 64, 12 * 64 + 8 * 4]
    # bijective over all (s,t) of a ragged table and inside the buffer
    rng = np.random.RandomState(0)
    lens = torch.from_numpy(rng.randint(0, 50, 200))
    off = layout.slice_row_offsets(lens)
    ss = torch.repeat_interleave(torch.arange(200), lens)
    tt = torch.cat([torch.arange(int(n)) for n in lens])
    e = layout.elem_index(off, ss, tt)
    assert e.unique().numel() == e.numel() and int(e.max()) < int(off[-1]) * 64
    assert layout.dense_rows(20000) == 20000 and layout.dense_rows(5) == 8


def test_shard_states_cover_and_align():
    for S in (1, 63, 64, 65, 1000, 65536, 2**20 + 5):
        for w in (1, 2, 4, 8):
            blocks = [layout.shard_states(S, w, r) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == S
            for (lo, hi), (lo2, _) in zip(blocks, blocks[1:]):
                assert hi == lo2 and lo <= hi and (lo % 64 == 0 or lo == hi)
