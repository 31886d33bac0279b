"""Build libdcarl_hip.so for gfx950 in-tree with hipcc (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["abi.hip", "trace.hip", "trace_final.hip", "trace_nwave_f32.hip", "trace_nwave_f64.hip", "bounds.hip", "buckets.hip", "ingest.hip", "comm.hip", "episodes.hip", "sampler.hip", "misc.hip", "rls.hip", "frenet.hip"]
LIB = os.path.join(HERE, "libdcarl_hip.so")
# -fno-honor-nans: keys built by integer bit-twiddling would otherwise be re-canonicalised (v_max_f64 x,x)
# before every v_max_f64; the path has no NaN semantics to preserve (DESIGN.md "NaN inputs").
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result", "-fno-honor-nans"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the DCARL HIP library cannot be built")
    return exe


def have_hipcc():
    return bool(shutil.which("hipcc")) or os.path.exists("/opt/rocm/bin/hipcc")


def source_id() -> str:
    """Content hash of everything the library is built from (mtimes do not survive a snapshot copy to the GPU box)."""
    import hashlib
    h = hashlib.sha256()
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)) + [os.path.join(HERE, "..", "include", "dcarl.h")]
    for d in deps:
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()[:16]


def built_id() -> str:
    try:
        with open(LIB + ".id") as f:
            return f.read().strip()
    except OSError:
        return ""


def needs_build():
    return not os.path.exists(LIB) or built_id() != source_id()


def build(force=False, verbose=False):
    """Compile and link in-tree.  Safe under concurrent callers (every rank of a torchrun launch sees a stale library at the
    same moment): an exclusive lock file serialises them, the late ones find the work done; the library and its id file
    appear by atomic rename, so nobody ever maps a half-written file."""
    if not force and not needs_build():
        return LIB
    import fcntl
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():          # another process built it while this one waited
                return LIB
            return _build_locked(verbose, force)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _object_key(src: str, sid: str) -> str:
    """What an object file depends on: its source, every header of csrc/ and include/ (any of them may be included), the flags —
    and, for abi.hip alone, the library's build id (it is the one file that embeds it)."""
    import hashlib
    h = hashlib.sha256()
    deps = [os.path.join(CSRC, src)] + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    deps.append(os.path.join(HERE, "..", "include", "dcarl.h"))
    for d in deps:
        with open(d, "rb") as f:
            h.update(os.path.basename(d).encode() + b"\0" + f.read())
    h.update(" ".join(FLAGS).encode())
    if src == "abi.hip":
        h.update(sid.encode())
    return h.hexdigest()[:16]


def _build_locked(verbose, force=False):
    objs = []
    sid = source_id()
    bdir = os.path.join(HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    procs = []
    for src in SOURCES:
        obj = os.path.join(bdir, src.replace(".hip", ".o"))
        objs.append(obj)
        key = _object_key(src, sid)
        try:
            with open(obj + ".key") as f:
                fresh = os.path.exists(obj) and f.read().strip() == key
        except OSError:
            fresh = False
        if fresh and not force:                           # incremental: an object whose inputs did not change is kept
            continue
        if os.path.exists(obj + ".key"):
            os.remove(obj + ".key")
        cmd = [hipcc(), *FLAGS, *([f'-DDCARL_BUILD_ID="{sid}"'] if src == "abi.hip" else []), "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, obj, key, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, obj, key, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
        with open(obj + ".key", "w") as f:
            f.write(key + "\n")
        if verbose and out:
            print(out.decode())
    tmp = f"{LIB}.tmp{os.getpid()}"
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-o", tmp]
    subprocess.check_call(cmd)
    with open(tmp + ".id", "w") as f:
        f.write(sid + "\n")
    if os.path.exists(LIB + ".id"):
        os.remove(LIB + ".id")                           # no moment at which a NEW library sits next to its OLD id
    os.replace(tmp, LIB)
    os.replace(tmp + ".id", LIB + ".id")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
