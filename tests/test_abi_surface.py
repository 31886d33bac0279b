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
    assert dcarl_amd.load_library().dcarl_version() == _lib.ABI_VERSION == 8
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
    # ABI version 8: the third trace, the top-2 census, host compaction + the packed group step
    assert lib.dcarl_true_step_values_f32(one, one, one, null, 4, 33, one, 1, 8, one, null) == -1 and b"A=33" in lib.dcarl_last_error()
    assert lib.dcarl_true_step_values_f32(one, one, one, null, 4, 11, one, 2, 8, one, null) == -1 and b"q_rows" in lib.dcarl_last_error()
    assert lib.dcarl_true_step_values_f64(one, one, one, null, 4, 11, one, 1, 8, C.c_void_p(8), null) == -1 and b"alignment" in lib.dcarl_last_error()
    assert lib.dcarl_true_step_values_f64(one, one, one, null, 0, 11, one, 1, 0, one, null) == 0
    assert lib.dcarl_top2_census_trace_f32(one, one, one, one, 4, 11, C.byref(p), null, null) == -1 and b"out" in lib.dcarl_last_error()
    assert lib.dcarl_top2_census_trace_f64(one, one, one, one, 0, 11, C.byref(p), one, null) == 0
    assert lib.dcarl_top2_census_table(null, 4, 11, C.byref(p), one, null) == -1
    assert lib.dcarl_top2_census_table(one, 4, 0, C.byref(p), one, null) == -1
    info = (C.c_int64 * 16)()
    assert lib.dcarl_host_compact_rows_f32(null, 5, 20, 11, one, info) == -1
    assert lib.dcarl_host_compact_rows_f32(one, 5, 70000, 11, one, info) == -1 and b"65536" in lib.dcarl_last_error()
    assert lib.dcarl_host_compact_rows_f32(null, 0, 20, 11, null, info) == 0 and info[8] == 0 and info[5] == -1
    ws256 = C.c_void_p(256)
    assert lib.dcarl_ingest_group_packed_f32(one, 5, 20, 11, 8, ws256, one, one, one, one, null, null) == -1           # NULL info
    assert lib.dcarl_ingest_group_packed_f32(one, 5, 20, 11, 0, ws256, one, one, one, one, one, null) == -1 and b"FORCE_DIRECT" in lib.dcarl_last_error()
    assert lib.dcarl_ingest_group_packed_f32(C.c_void_p(4), 5, 20, 11, 8, ws256, one, one, one, one, one, null) == -1
    assert lib.dcarl_ingest_group_packed_f32(one, 0, 20, 11, 8, ws256, one, one, one, one, one, null) == -1

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


# ---- the host side of the C-ABI on a box without a GPU: sizing arithmetic and launch plans (run again under ASan + UBSan by
#      tests/test_abi_host_sanitized.py) -------------------------------------------------------------------------------------
def _no_gpu():
    import torch
    return not torch.cuda.is_available()


def test_workspace_sizing_sweep_without_gpu():
    """Every sizing function over the whole admissible range of its arguments (and past it): never negative, never less than the records
    themselves, 0 for what the header calls invalid — the workspace-layout arithmetic is 64-bit host code with no GPU in it, which is
    where round 3's off-by-one lived (signed overflow shows up under UBSan in the sanitized run)."""
    lib = dcarl_amd.load_library()
    Ns = [0, 1, 3, 4, 5, 63, 64, 65, 6655, 6656, 6657, 2 ** 20 - 1, 2 ** 20, 2 ** 20 + 1, 2 ** 24 + 7, 2 ** 31 - 1]
    Ss = [1, 2, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 65535, 65536, 65537, 2 ** 24, 2 ** 31 - 1]
    for S in Ss:
        assert lib.dcarl_slot_order_workspace_bytes(S) > 0
        for A in (1, 11, 16, 17, 32):
            for vb in (4, 8):
                for flags in range(16):
                    for buckets in (0, 1):
                        for N in Ns:
                            b = lib.dcarl_ingest_workspace_bytes(N, S, A, vb, flags, buckets)
                            assert b >= 8 * N, (N, S, A, vb, flags, buckets, b)        # at least the compact records themselves
    assert lib.dcarl_slot_order_workspace_bytes(0) == 0 and lib.dcarl_slot_order_workspace_bytes(-5) == 0
    for bad in ((-1, 4, 11, 4, 0, 0), (5, 0, 11, 4, 0, 0), (5, 4, 0, 4, 0, 0), (5, 4, 33, 4, 0, 0), (5, 4, 11, 3, 0, 0)):
        assert lib.dcarl_ingest_workspace_bytes(*bad) == 0, bad
    for kind in range(0, 9):
        for S in (-1, 0, 1, 1000, 2 ** 31 - 1, 2 ** 40):
            for N in (-1, 0, 1, 5000, 2 ** 31 - 1, 2 ** 40):
                for A in (0, 1, 11, 32):
                    assert lib.dcarl_workspace_bytes(kind, S, A, N) >= 0
    for N in (-3, 0, 1, 255, 256, 257, 2 ** 20, 2 ** 31, 2 ** 40):
        assert lib.dcarl_scan_workspace_bytes(N) >= 0
        for Q in (0, 1, 64, 8192, 2 ** 31 - 1):
            assert lib.dcarl_rls_workspace_bytes(N, Q) >= 0


@pytest.mark.skipif(not _no_gpu(), reason="fake device addresses: only where no launch can execute")
def test_launch_plans_with_fake_device_addresses_without_gpu():
    """Every launcher's HOST side — dispatch, plan and grid arithmetic, the ingest's workspace layout, the stamp ring — run on
    well-aligned fake device addresses on a box where no kernel can execute (the HIP launch itself fails with 'no device'): each call
    returns an error code or DCARL_OK for an empty shape and nothing crashes.  Nothing is dereferenced on the host: the addresses
    are never mapped."""
    lib = dcarl_amd.load_library()
    p = dcarl_amd.Params().to_c()
    base = 0x7f0000000000
    ptrs = [C.c_void_p(base + (i << 32)) for i in range(16)]
    a, b, c, d, e, f, g, h, i_, j, k, l_, m, n, o, q = ptrs
    null = C.c_void_p(None)
    st = _lib.CTraceState(*[x.value for x in ptrs[:6]])
    rcs = []
    for S in (1, 64, 65, 300, 4096, 65536, 70000):
        for A in (1, 5, 11, 12, 13, 16, 17, 24, 32):
            for steps in (True, False):
                sv, sa = (g, h) if steps else (null, null)
                rcs.append(lib.dcarl_trace_f32(a, b, c, d, null, S, A, C.byref(p), sv, sa, i_, j, k, l_, m, null))
                rcs.append(lib.dcarl_trace_f64(a, b, c, d, e, S, A, C.byref(p), sv, sa, i_, j, k, l_, m, null))
                rcs.append(lib.dcarl_trace_f32(a, b, c, d, null, S, A, C.byref(p), null, null, null, j, k, l_, m, null))    # the final table
                rcs.append(lib.dcarl_trace_resume_f32(a, b, c, d, e, S, A, C.byref(p), C.byref(st), 1, sv, sa, l_, m, null))
                rcs.append(lib.dcarl_trace_resume_f64(a, b, c, d, null, S, A, C.byref(p), C.byref(st), 0, sv, sa, l_, m, null))
            for nd, hint in ((0, 3), (0, 91), (0, 1818), (64, 64), (7, 0)):
                seg = null if nd else b
                rcs.append(lib.dcarl_bounds_csr_f32(a, seg, nd, hint, S, A, C.byref(p), c, d, e, f, null))
                rcs.append(lib.dcarl_bounds_csr_f64(a, seg, nd, hint, S, A, C.byref(p), c, d, e, f, null))
    for N in (0, 1, 5, 6655, 6656, 6657, 70001, 2 ** 20, 2 ** 20 + 77, 6656 * 128 * 3 + 1, 2 ** 27 + 3, 2 ** 31 - 1):
        for S, A in ((1, 30), (20, 11), (300, 32), (2048, 11), (5000, 11), (65536, 16), (70000, 11), (2 ** 24, 11)):
            for flags in range(16):
                for fn_g, fn_p in ((lib.dcarl_ingest_group_f32, lib.dcarl_ingest_pack_f32), (lib.dcarl_ingest_group_f64, lib.dcarl_ingest_pack_f64)):
                    rcs.append(fn_g(a, N, S, A, flags, b, c, d, e, f, g, h, null))
                    bands = N // 32 + 2 * ((S + 63) // 64) + 2
                    rcs.append(fn_p(N, S, A, flags, b, c, d, f, bands, i_, j, k, l_, null))
                    rcs.append(fn_p(N, S, A, flags ^ 1, b, c, d, f, bands, i_, j, k, l_, null))      # other flags than the group call's: refused
                    assert rcs[-1] != 0 or N > 0x7fffffff
                rcs.append(lib.dcarl_ingest_group_pairs_f32(a, b, c, N, S, A, flags, d, e, f, g, h, i_, null))
            rcs.append(lib.dcarl_ingest_buckets_f32(a, N, S, A, b, c, d, e, null))
            rcs.append(lib.dcarl_ingest_buckets_f64(a, N, S, A, b, c, d, e, null))
            rcs.append(lib.dcarl_slot_order(a, S, 20000, 1, b, c, d, e, f, g, null))
            rcs.append(lib.dcarl_export_records_f32(a, b, c, d, null, S, N // max(S, 1), 1, null, null, (N // max(S, 1)) * S, e, null))
    for i in range(200):                                     # the stamp ring (64 slots) wraps; an evicted workspace is packed unchecked
        ws = C.c_void_p(base + 0x1000 * (i + 1))
        lib.dcarl_ingest_group_f32(a, 1000 + i, 20, 11, 0, ws, c, d, e, f, g, h, null)
    for N in (0, 1, 12, 10 ** 6, 2 ** 30, 2 ** 33):
        rcs.append(lib.dcarl_sample_pairs(a, 20, 11, N, 50.0, 0, 0, 1, b, c, d, null, null))
        rcs.append(lib.dcarl_sample_pairs(a, 65536, 11, N, 50.0, 0, 7, 1, b, c, d, e, null))
        rcs.append(lib.dcarl_scan_f64(a, b, N, c, null))
        rcs.append(lib.dcarl_count_nonfinite(a, 4, N, b, null))
        rcs.append(lib.dcarl_sample_state_records(a, 4096, 11, 11, max(4, N // 4096 // 4 * 4), 50.0, 1, 0, b, c, null))
        rcs.append(lib.dcarl_state_ids(a, null, min(N, 2 ** 31 - 1), 20, 0, b, c, d, null))
        rcs.append(lib.dcarl_index_states_f64(a, min(N, 2 ** 31 - 1), 20, b, 0, c, d, e, f, null))
        rcs.append(lib.dcarl_episode_returns_f64(a, b, c, d, N, e, f, g, null))
    assert all(isinstance(r, int) for r in rcs)
    assert lib.dcarl_trace_status(null) != 0                 # no device: the status read must say so, not claim a clean run


def pack_rows_numpy(rows, S, A):
    """The rule of dcarl_host_compact_rows_f32 (== the device row ingest's convert()) in NumPy: -> (packed uint64, info words 3..8)."""
    import numpy as np
    sd, ad, wd = rows[:, 0], rows[:, 2], rows[:, 3]
    def ids(x):
        nf = ~np.isfinite(x)
        big = np.abs(x) >= 2.0e9
        with np.errstate(invalid="ignore"):
            i = np.where(nf, np.iinfo(np.int32).min, np.where(big, np.where(x < 0, np.iinfo(np.int32).min, np.iinfo(np.int32).max),
                                                              np.trunc(np.where(nf | big, 0.0, x)))).astype(np.int64)
        return i, nf
    si, s_nf = ids(sd)
    ai, a_nf = ids(ad)
    flags = (2 if (s_nf.any() or a_nf.any()) else 0) | (1 if (~(np.abs(wd) <= 3.4028234663852886e38)).any() else 0)
    st = np.where((si >= 0) & (si < S), si, 0).astype(np.uint64)
    a = np.where((ai >= 0) & (ai < A), ai, 0).astype(np.uint64)
    with np.errstate(over="ignore", invalid="ignore"):
        wb = wd.astype(np.float32).view(np.uint32).astype(np.uint64)
    return (st << np.uint64(5)) | a | (wb << np.uint64(32)), [int(ai.max()), int(si.min()), int(si.max()), int(ai.min()), flags, len(rows)]


def test_argument_validation_without_gpu_host_compaction_matches_its_rule():
    """dcarl_host_compact_rows_f32 is plain host code: it runs here (and under ASan + UBSan, tests/test_abi_host_sanitized.py) against a
    NumPy statement of the row ingest's rule — ids truncated toward zero, range and NaN / Inf handling, f32 rounding of the reward —
    on clean rows, on every kind of offending row, in one call and cut into ranges."""
    import numpy as np
    lib = dcarl_amd.load_library()
    rng = np.random.RandomState(5)
    n, S, A = 10007, 300, 11
    rows = np.stack([rng.randint(0, S, n) + rng.rand(n) * 0.9, rng.rand(n), rng.randint(0, A, n) + rng.rand(n) * 0.9,
                     rng.uniform(-50, 100, n) + 50 * rng.standard_normal(n)], 1)
    bad = rows.copy()
    bad[7, 0], bad[8, 0], bad[9, 2], bad[10, 2] = -1.5, 3.0e9, 11.0, -0.5            # -1.5 truncates to -1: out of range; -0.5 to 0: fine
    bad[11, 3], bad[12, 3], bad[13, 0], bad[14, 3] = np.nan, 1e39, np.inf, -np.inf
    for name, r in (("clean", rows), ("bad", bad)):
        out = np.empty(n, np.uint64)
        info = (C.c_int64 * 16)()
        assert lib.dcarl_host_compact_rows_f32(C.c_void_p(r.ctypes.data), n, S, A, C.c_void_p(out.ctypes.data), info) == 0
        want, winfo = pack_rows_numpy(r, S, A)
        assert np.array_equal(out, want), name
        assert [info[k] for k in (3, 4, 5, 6, 7, 8)] == winfo, (name, list(info), winfo)
    assert winfo[4] == 3 and winfo[1] < 0 and winfo[2] > S
    # the Python wrapper: ranges over a thread pool, info words combined, the ingest's own errors from them
    from concurrent.futures import ThreadPoolExecutor
    from dcarl_amd.records import check_ingest_info, compact_rows_host
    big = np.tile(rows, (8, 1))
    with ThreadPoolExecutor(4) as pool:
        rec, h = compact_rows_host(big, S, A, pool=pool, pieces=4)
    want, winfo = pack_rows_numpy(big, S, A)
    assert np.array_equal(rec.view(np.uint64), want) and h[3:9] == winfo
    check_ingest_info([0, 0, 0] + h[3:], S, A, len(big))
    import pytest
    _, hb = compact_rows_host(bad, S, A)
    with pytest.raises(ValueError):
        check_ingest_info([0, 0, 0] + hb[3:], S, A, n)
    b2 = rows.copy()
    b2[5, 0] = -3
    with pytest.raises(IndexError):
        check_ingest_info([0, 0, 0] + compact_rows_host(b2, S, A)[1][3:], S, A, n)
