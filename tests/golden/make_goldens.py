#!/usr/bin/env python3
"""Generate the golden fixtures by RUNNING THE UNMODIFIED REFERENCE (build container only).

Usage (from the repo root, needs /root/reference):  python tests/golden/make_goldens.py

Writes into tests/golden/:
  bounds_random.npz   the four bound functions (S1:10-28) on ~1000 random buckets
  sim1_trace.npz      script globals of Simulation_1/test_DCARL.py on the bundled data
  sim2_trace.npz      script globals of Simulation_2/test_DCARL.py on the bundled data
  sampler_seed{0,1,2}.npz  seeded Data_Generation() outputs + the raw noise streams
  refused_inputs.npz  what the reference does with the two kinds of input this library REFUSES (negative state ids, which
                      Python's indexing wraps; a NaN reward, which np.argmax picks): Simulation_2/test_DCARL.py run on two
                      small tables, inputs and script globals
and copies the reference's bundled DATA files (.npy record tables; data, not code)
to the same relative paths the drop-in scripts read them from.

Nothing from /root/reference is imported at test time: tests only read the .npz/.npy.
"""
import importlib.util
import io
import contextlib
import os
import random
import runpy
import shutil
import sys
import tempfile

import numpy as np

REF = os.environ.get("DCARL_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
S1 = "Simulation_testing/Simulation_1/test_DCARL.py"
S2 = "Simulation_testing/Simulation_2/test_DCARL.py"
DS_DIR = "Simulation_testing/Simulation_Data_Collection/Data_Sampling"

os.environ["MPLBACKEND"] = "Agg"


def load_functions(path):
    spec = importlib.util.spec_from_file_location("ref_mod", os.path.join(REF, path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def ragged(lists, dtype):
    flat = np.concatenate([np.asarray(l, dtype=dtype) for l in lists]) if lists else np.zeros(0, dtype)
    off = np.zeros(len(lists) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(l) for l in lists])
    return flat, off


def gen_bounds():
    m = load_functions(S2)
    rng = np.random.RandomState(20260928)
    sizes = list(range(11, 75)) + [100, 128, 255, 256, 257, 500, 1000, 1917, 2370, 4096]
    xs, off, ub, lb, ci, mv = [], [0], [], [], [], []
    for sigma in (0.2, 50.0):
        for n in sizes:
            for rep in range(7):
                mu = rng.uniform(-50, 100)
                x = mu + sigma * rng.standard_normal(n)
                xs.append(x)
                off.append(off[-1] + n)
                ub.append(m.upper_bound(x)); lb.append(m.lower_bound(x))
                ci.append(m.CI_lower_bound(x)); mv.append(m.mean_value(x))
    # degenerate buckets: constant, two-valued, huge values
    for x in (np.full(11, 7.25), np.full(64, -50.0), np.array([0.0, 1.0] * 8), np.full(20, 1e6),
              np.linspace(-50, 100, 33)):
        xs.append(x); off.append(off[-1] + len(x))
        ub.append(m.upper_bound(x)); lb.append(m.lower_bound(x))
        ci.append(m.CI_lower_bound(x)); mv.append(m.mean_value(x))
    # non-default alpha/scale
    extra = []
    for alpha, scale in ((0.01, 150), (0.1, 100), (0.05, 1.0)):
        x = xs[5]
        extra.append([alpha, scale, m.upper_bound(x, alpha, -50, scale), m.lower_bound(x, alpha, -50, scale),
                      m.CI_lower_bound(x, alpha, -50, scale)])
    np.savez_compressed(os.path.join(HERE, "bounds_random.npz"), x=np.concatenate(xs),
                        off=np.array(off, dtype=np.int64), upper=np.array(ub), lower=np.array(lb),
                        ci_lower=np.array(ci), mean_value=np.array(mv), extra=np.array(extra))
    print("bounds_random:", len(ub), "buckets")


def run_script(path):
    cwd = os.getcwd()
    os.chdir(REF)
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            g = runpy.run_path(os.path.join(REF, path), run_name="__main__")
    finally:
        os.chdir(cwd)
    return g, buf.getvalue()


def gen_sim(path, name, with_overall):
    g, out = run_script(path)
    sv, sv_off = ragged(g["step_TSRL_value"], np.float64)
    sa, _ = ragged(g["step_TSRL_act"], np.int64)
    tv, _ = ragged(g["true_step_TSRL_value"], np.float64)
    bl = np.array([[len(b) for b in row] for row in g["data_state_act"]], dtype=np.int64)
    d = dict(step_value=sv, step_off=sv_off, step_act=sa, true_step_value=tv,
             TSRL_value=np.array(g["TSRL_value"], dtype=np.float64),
             activation_step=np.asarray(g["activation_step"], dtype=np.int64),
             bucket_len=bl, stdout=np.array(out))
    if with_overall:
        d["overall_value"] = np.asarray(g["overall_value"], dtype=np.float64)
        d["sorted_state_data_len"] = np.asarray(g["sorted_state_data_len"], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, name), **d)
    print(name, "activation_step", d["activation_step"], "stdout tail:", (out.strip().splitlines() or [""])[-1])


def gen_sampler(seed):
    sys.path.insert(0, os.path.join(REF, DS_DIR))
    import warnings
    warnings.simplefilter("ignore", DeprecationWarning)
    ds = load_functions(os.path.join(DS_DIR, "data_sampling.py"))
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "Simulation_testing/Simulation_Data_Collection"))
        os.chdir(tmp)
        try:
            np.random.seed(seed); random.seed(seed)
            ds.Data_Generation()
            base = "Simulation_testing/Simulation_Data_Collection/"
            data = np.load(base + "data.npy"); q = np.load(base + "action_value.npy")
            states = np.load(base + "states.npy")
        finally:
            os.chdir(cwd)
    # capture the raw streams by replaying the same legacy generators (draw order of DS:39-55)
    rs = np.random.RandomState(seed); random.seed(seed)
    u_states = rs.random_sample(20)
    u_q = np.stack([rs.random_sample(11) for _ in range(20)])
    z_visit = rs.standard_normal(50000)
    idxs = np.floor((3.0 + z_visit) / 6 * 20).astype(int)
    acts, zs = [], []
    for idx in idxs:
        if idx < 0 or idx >= 20:
            continue
        acts.append(random.randint(0, 10)); zs.append(rs.standard_normal(1)[0])
    # helper functions on fixed inputs
    np.random.seed(seed + 100)
    rsn = ds.random_state_norm(20, 1000)
    rs2 = np.random.RandomState(seed + 100)
    z_rsn = rs2.standard_normal(1000)
    # random_state_manual (DS:19-28; defined, never called by Data_Generation): its output and the Python-random streams
    # behind it, captured by replaying the same calls in the same order
    random.seed(seed + 200)
    rsm = ds.random_state_manual(20, 1000)
    random.seed(seed + 200)
    m_u, m_r = [], []
    for _ in range(1000):
        x = random.random()
        m_u.append(x)
        if x > 0.1:
            m_r.append(random.randint(1, 19))
    np.savez_compressed(os.path.join(HERE, f"sampler_seed{seed}.npz"), data=data, action_value=q,
                        random_state_manual_out=np.asarray(rsm, dtype=np.int64), random_state_manual_u=np.array(m_u),
                        random_state_manual_r=np.array(m_r, dtype=np.int64),
                        states=states, u_states=u_states, u_q=u_q, z_visit=z_visit,
                        acts=np.array(acts, dtype=np.int64), z_reward=np.array(zs),
                        random_state_norm_out=np.asarray(rsn, dtype=np.int64), random_state_norm_z=z_rsn)
    print(f"sampler seed {seed}: rows {data.shape}")


def run_script_on(path, files):
    """The unmodified script run in a scratch directory that holds `files` (relative path -> array) where it expects its inputs."""
    cwd = os.getcwd()
    buf = io.StringIO()
    with tempfile.TemporaryDirectory() as tmp:
        for rel_path, arr in files.items():
            os.makedirs(os.path.dirname(os.path.join(tmp, rel_path)), exist_ok=True)
            np.save(os.path.join(tmp, rel_path), arr)
        os.chdir(tmp)
        try:
            import warnings
            with contextlib.redirect_stdout(buf), warnings.catch_warnings():
                warnings.simplefilter("ignore")
                g = runpy.run_path(os.path.join(REF, path), run_name="__main__")
        finally:
            os.chdir(cwd)
    return g


def gen_refused():
    """Two behaviours of the reference this library refuses at its boundary instead of reproducing (include/dcarl.h): S2:77-80 indexes
    data_state_act[idx] with whatever int(row[0]) is — a NEGATIVE id wraps (Python indexing: -1 is state 19 of 20) — and S2:92
    takes np.argmax over a table that may hold NaN — the first NaN wins.  Recorded here from the unmodified script so that the
    divergence is a golden, not a header comment."""
    rng = np.random.RandomState(20260930)
    q = rng.uniform(-50, 100, (20, 11))
    states = rng.uniform(0, 1, 20)
    out = {}
    for kind in ("negative_id", "nan_reward"):
        n = 600
        idx = rng.randint(0, 3, n)
        act = rng.randint(0, 11, n)
        R = q[idx, act] + 50.0 * rng.standard_normal(n)
        if kind == "negative_id":
            neg = rng.rand(n) < 0.25
            idx = np.where(neg, np.where(rng.rand(n) < 0.5, -1, -20), idx)         # -1 -> state 19, -20 -> state 0
            R = q[idx, act] + 50.0 * rng.standard_normal(n)
        else:
            a_big = 1 + int(np.argmax([((idx == 1) & (act == a)).sum() for a in range(1, 11)]))   # state 1's largest non-rule bucket
            k = int(np.flatnonzero((idx == 1) & (act == a_big))[4])                # its 5th sample: the bucket's value is NaN from n = 11 on
            out["nan_reward_bucket"] = np.array([1, a_big])
            R[k] = np.nan
        data = np.stack([idx.astype(np.float64), states[idx], act.astype(np.float64), R], 1)
        g = run_script_on(S2, {"Simulation_testing/Simulation_2/data.npy": data,
                               "Simulation_testing/Simulation_2/action_value.npy": q})
        sv, sv_off = ragged(g["step_TSRL_value"], np.float64)
        sa, _ = ragged(g["step_TSRL_act"], np.int64)
        out.update({f"{kind}_data": data, f"{kind}_step_value": sv, f"{kind}_step_off": sv_off, f"{kind}_step_act": sa,
                    f"{kind}_TSRL_value": np.array(g["TSRL_value"], dtype=np.float64),
                    f"{kind}_activation_step": np.asarray(g["activation_step"], dtype=np.int64),
                    f"{kind}_bucket_len": np.array([[len(b) for b in row] for row in g["data_state_act"]], dtype=np.int64)})
    out["action_value"] = q
    np.savez_compressed(os.path.join(HERE, "refused_inputs.npz"), **out)
    print("refused_inputs.npz: records filed under wrapped states",
          out["negative_id_bucket_len"].sum(1)[[0, 19]], "; NaN steps", int(np.isnan(out["nan_reward_step_value"]).sum()))


def copy_data():
    pairs = [("Simulation_testing/Simulation_1/data_carla.npy",) * 2,
             ("Simulation_testing/Simulation_1/action_value_carla.npy",) * 2,
             ("Simulation_testing/Simulation_1/states_carla.npy",) * 2,
             ("Simulation_testing/Simulation_2/data.npy",) * 2,
             ("Simulation_testing/Simulation_2/action_value.npy",) * 2,
             ("Simulation_testing/Simulation_2/states.npy",) * 2]
    for src, dst in pairs:
        os.makedirs(os.path.dirname(os.path.join(REPO, dst)), exist_ok=True)
        shutil.copyfile(os.path.join(REF, src), os.path.join(REPO, dst))
        os.chmod(os.path.join(REPO, dst), 0o644)
    print("copied bundled data tables")


if __name__ == "__main__":
    gen_bounds()
    gen_sim(S1, "sim1_trace.npz", False)
    gen_sim(S2, "sim2_trace.npz", True)
    for sd in (0, 1, 2):
        gen_sampler(sd)
    gen_refused()
    copy_data()
