"""ctypes wrapper around oracle/_build/libdcarl_oracle.so (TEST INFRASTRUCTURE ONLY)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libdcarl_oracle.so")


class OrcParams(C.Structure):
    _fields_ = [("rule_act", C.c_int32), ("n_thres", C.c_int32), ("alpha", C.c_double), ("scale", C.c_double),
                ("cap", C.c_double), ("init_rule", C.c_double), ("init_other", C.c_double)]


def build(force=False):
    src = os.path.join(_HERE, "dcarl_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_max_threads.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def params(rule_act=0, n_thres=10, alpha=0.05, scale=150.0, cap=100.0, init_rule=100.0, init_other=-50.0):
    return OrcParams(rule_act, n_thres, alpha, scale, cap, init_rule, init_other)


def max_threads():
    return lib().orc_max_threads()


def set_threads(n):
    lib().orc_set_threads(int(n))


def trace(R, act, state_off, S, A, p=None, recompute=False, want_steps=True):
    """Records grouped by state.  R float32 or float64 1-D, act uint8, state_off int64[S+1]."""
    p = p or params()
    R = np.ascontiguousarray(R)
    assert R.dtype in (np.float32, np.float64)
    act = np.ascontiguousarray(act, dtype=np.uint8)
    state_off = np.ascontiguousarray(state_off, dtype=np.int64)
    N = int(state_off[-1])
    sv = np.empty(N, np.float64) if want_steps else None
    sa = np.empty(N, np.uint8) if want_steps else None
    latch = np.empty(S, np.int32)
    if recompute:
        lib().orc_trace_recompute(_p(R), int(R.dtype == np.float32), _p(act), _p(state_off), S, A, C.byref(p),
                                  _p(sv), _p(sa), _p(latch))
        return dict(step_val=sv, step_act=sa, activation_step=latch)
    V = np.empty((S, A), np.float64)
    n = np.empty((S, A), np.int32)
    vmax = np.empty(S, np.float32)
    amax = np.empty(S, np.int32)
    lib().orc_trace(_p(R), int(R.dtype == np.float32), _p(act), _p(state_off), S, A, C.byref(p), _p(sv), _p(sa),
                    _p(latch), _p(V), _p(n), _p(vmax), _p(amax))
    return dict(step_val=sv, step_act=sa, activation_step=latch, V=V, n=n, vmax=vmax, amax=amax)


def bounds_csr(values, seg_off, S, A, p=None):
    p = p or params()
    values = np.ascontiguousarray(values)
    assert values.dtype in (np.float32, np.float64)
    seg_off = np.ascontiguousarray(seg_off, dtype=np.int64)
    V = np.empty((S, A), np.float64)
    n = np.empty((S, A), np.int32)
    vmax = np.empty(S, np.float32)
    amax = np.empty(S, np.int32)
    lib().orc_bounds_csr(_p(values), int(values.dtype == np.float32), _p(seg_off), S, A, C.byref(p), _p(V), _p(n),
                         _p(vmax), _p(amax))
    return dict(V=V, n=n, vmax=vmax, amax=amax)


def overall(step_val, act_step, state_off, rec_state, rec_pos):
    step_val = np.ascontiguousarray(step_val, dtype=np.float64)
    act_step = np.ascontiguousarray(act_step, dtype=np.int32)
    state_off = np.ascontiguousarray(state_off, dtype=np.int64)
    rec_state = np.ascontiguousarray(rec_state, dtype=np.int32)
    rec_pos = np.ascontiguousarray(rec_pos, dtype=np.int64)
    N = len(rec_state)
    out = np.empty(N, np.float64)
    lib().orc_overall(_p(step_val), _p(act_step), _p(state_off), _p(rec_state), _p(rec_pos), C.c_int64(N),
                      len(act_step), _p(out))
    return out


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return tuple(int(v) for v in o)


def sample_state_records(Q, T, seed, stream=0, sigma=50.0):
    Q = np.ascontiguousarray(Q, dtype=np.float64)
    S, A = Q.shape
    act = np.empty((S, T), np.uint8)
    R = np.empty((S, T), np.float64)
    lib().orc_sample_state_records(_p(Q), S, A, C.c_int64(T), C.c_uint64(seed), C.c_uint32(stream), C.c_double(sigma),
                                   _p(act), _p(R))
    return act, R


def sample_pairs(Q, N, seed, offset=0, stream=1, sigma=50.0, want_z=False):
    Q = np.ascontiguousarray(Q, dtype=np.float64)
    S, A = Q.shape
    idx = np.empty(N, np.int32)
    act = np.empty(N, np.int32)
    R = np.empty(N, np.float64)
    z = np.empty(N, np.float64) if want_z else None
    lib().orc_sample_pairs(_p(Q), S, A, C.c_int64(N), C.c_uint64(seed), C.c_uint64(offset), C.c_uint32(stream),
                           C.c_double(sigma), _p(idx), _p(act), _p(R), _p(z))
    return (idx, act, R, z) if want_z else (idx, act, R)


def sample_state_records_ragged(Q, lengths, seed, stream=0, sigma=50.0, n_live=None):
    """State-major (act uint8, R float64, state_off) of dcarl_sample_state_records_ragged."""
    Q = np.ascontiguousarray(np.atleast_2d(Q), dtype=np.float64)
    lengths = np.asarray(lengths, dtype=np.int64)
    S = len(lengths)
    q_rows, A = Q.shape
    off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    act = np.empty(int(off[-1]), np.uint8)
    R = np.empty(int(off[-1]), np.float64)
    nl = None if n_live is None else np.ascontiguousarray(n_live, dtype=np.int32)
    lib().orc_sample_state_records_ragged(_p(Q), q_rows, S, A, _p(off), _p(nl), C.c_uint64(seed), C.c_uint32(stream),
                                          C.c_double(sigma), _p(act), _p(R))
    return act, R, off


def sample_buckets(Q, seg_off, S, seed, stream=2, sigma=50.0):
    Q = np.ascontiguousarray(np.atleast_2d(Q), dtype=np.float64)
    q_rows, A = Q.shape
    seg_off = np.ascontiguousarray(seg_off, dtype=np.int64)
    values = np.empty(int(seg_off[-1]), np.float64)
    lib().orc_sample_buckets(_p(Q), q_rows, S, A, _p(seg_off), C.c_uint64(seed), C.c_uint32(stream), C.c_double(sigma),
                             _p(values))
    return values
