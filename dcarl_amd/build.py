"""Build libdcarl_hip.so for gfx950 in-tree with hipcc (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["abi.hip", "trace.hip", "trace_final.hip", "diag.hip", "trace_nwave_f32.hip", "trace_nwave_f64.hip", "bounds.hip", "buckets.hip", "ingest.hip", "comm.hip", "episodes.hip", "sampler.hip", "misc.hip", "rls.hip", "frenet.hip"]
LIB = os.path.join(HERE, "libdcarl_hip.so")
# Variants of the one source tree.  "" = the product library (no environment knobs, no measurement-only kernel instances).
# "ab" = -DDCARL_AB_BUILD: the launchers' choices can be overridden through DCARL_* environment variables and the two- / four-wave,
# unfenced and small single-wave instances of the online kernel exist — what the tests of every kernel instance and tools/'s A/B
# scripts load (dcarl_amd._lib.use_variant("ab")); never what a caller of the package gets.
VARIANT_FLAGS = {"": [], "ab": ["-DDCARL_AB_BUILD"]}
# -fno-honor-nans: keys built by integer bit-twiddling would otherwise be re-canonicalised (v_max_f64 x,x)
# before every v_max_f64; the path has no NaN semantics to preserve (DESIGN.md "NaN inputs").
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result", "-fno-honor-nans"]
# per translation unit.  The f32 online kernel with the backend's max-ILP scheduling strategy: same box, the three online workloads
# 3.062 -> 3.033, 2.549 -> 2.523, 1.179 -> 1.171 ms (round 6, profiles/r06_ab_online_sched_strategy.txt); no instance spills with it.
SOURCE_FLAGS = {"trace_nwave_f32.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the DCARL HIP library cannot be built")
    return exe


def have_hipcc():
    return bool(shutil.which("hipcc")) or os.path.exists("/opt/rocm/bin/hipcc")


def lib_path(variant: str = "") -> str:
    return LIB if not variant else os.path.join(HERE, f"libdcarl_hip_{variant}.so")


_TOOLCHAIN = None


def toolchain_id() -> str:
    """The compiler's own identity (`hipcc --version`): objects of another ROCm must not survive in the cache."""
    global _TOOLCHAIN
    if _TOOLCHAIN is None:
        try:
            _TOOLCHAIN = subprocess.run([hipcc(), "--version"], capture_output=True, text=True, timeout=120).stdout.strip()
        except Exception:   # noqa: BLE001
            _TOOLCHAIN = "unknown"
    return _TOOLCHAIN


def source_id(variant: str = "") -> str:
    """Content hash of everything the library is built from (mtimes do not survive a snapshot copy to the GPU box); a variant's
    id carries its name ("<hash>+ab"), which is also what dcarl_build_id() of that library returns."""
    import hashlib
    h = hashlib.sha256()
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)) + [os.path.join(HERE, "..", "include", "dcarl.h")]
    for d in deps:
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS + VARIANT_FLAGS[variant] + [f"{k}:{' '.join(v)}" for k, v in sorted(SOURCE_FLAGS.items())]).encode())
    return h.hexdigest()[:16] + (f"+{variant}" if variant else "")


def built_id(variant: str = "") -> str:
    try:
        with open(lib_path(variant) + ".id") as f:
            return f.read().strip()
    except OSError:
        return ""


def needs_build(variant: str = ""):
    return not os.path.exists(lib_path(variant)) or built_id(variant) != source_id(variant)


def build(force=False, verbose=False, variant: str = ""):
    """Compile and link in-tree.  Safe under concurrent callers (every rank of a torchrun launch sees a stale library at the
    same moment): an exclusive lock file serialises them, the late ones find the work done; the library and its id file
    appear by atomic rename, so nobody ever maps a half-written file."""
    lib = lib_path(variant)
    if not force and not needs_build(variant):
        return lib
    import fcntl
    with open(lib + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build(variant):   # another process built it while this one waited
                return lib
            return _build_locked(verbose, force, variant)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _object_key(src: str, sid: str, variant: str = "") -> str:
    """What an object file depends on: its source, every header of csrc/ and include/ (any of them may be included), the flags —
    and, for abi.hip alone, the library's build id (it is the one file that embeds it)."""
    import hashlib
    h = hashlib.sha256()
    deps = [os.path.join(CSRC, src)] + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    deps.append(os.path.join(HERE, "..", "include", "dcarl.h"))
    for d in deps:
        with open(d, "rb") as f:
            h.update(os.path.basename(d).encode() + b"\0" + f.read())
    h.update(" ".join(FLAGS + VARIANT_FLAGS[variant] + SOURCE_FLAGS.get(src, [])).encode())
    h.update(toolchain_id().encode())
    if src == "abi.hip":
        h.update(sid.encode())
    return h.hexdigest()[:16]


def build_info(variant: str = "") -> dict:
    """What the last build of this variant did (written next to the library): forced or incremental, which objects were compiled and
    which re-used, seconds, compiler — so that a driver's "does it build" check can show that it really compiled."""
    import json
    try:
        with open(lib_path(variant) + ".buildinfo") as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def _build_locked(verbose, force=False, variant: str = ""):
    import json
    import time
    t_begin = time.time()
    compiled, reused = [], []
    objs = []
    sid = source_id(variant)
    LIB = lib_path(variant)                                # noqa: N806  (shadows the module's product path on purpose)
    bdir = os.path.join(HERE, "build" + (f"_{variant}" if variant else ""))
    os.makedirs(bdir, exist_ok=True)
    procs = []
    for src in SOURCES:
        obj = os.path.join(bdir, src.replace(".hip", ".o"))
        objs.append(obj)
        key = _object_key(src, sid, variant)
        try:
            with open(obj + ".key") as f:
                fresh = os.path.exists(obj) and f.read().strip() == key
        except OSError:
            fresh = False
        if fresh and not force:                           # incremental: an object whose inputs did not change is kept
            reused.append(src)
            continue
        compiled.append(src)
        if os.path.exists(obj + ".key"):
            os.remove(obj + ".key")
        cmd = [hipcc(), *FLAGS, *VARIANT_FLAGS[variant], *SOURCE_FLAGS.get(src, []), *([f'-DDCARL_BUILD_ID="{sid}"'] if src == "abi.hip" else []), "-c",
               os.path.join(CSRC, src), "-o", obj]
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
    with open(LIB + ".buildinfo", "w") as f:
        json.dump(dict(library=os.path.basename(LIB), variant=variant or "product", build_id=sid, build_mode="forced (every object compiled)" if force
                       else "incremental", objects_compiled=compiled, objects_reused=reused, seconds=round(time.time() - t_begin, 1),
                       flags=FLAGS + VARIANT_FLAGS[variant], hipcc=toolchain_id().splitlines()[0] if toolchain_id() else ""), f, indent=1)
    return LIB


def build_all(force=False, verbose=False, variants=("", "ab")):
    """Every variant, their compilations side by side (the critical path of either is ONE translation unit, the f32 online
    kernel: two variants one after the other would double the wall time for nothing)."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(len(variants)) as ex:
        return list(ex.map(lambda v: build(force=force, verbose=verbose, variant=v), variants))


def build_host_sanitized(out_dir: str | None = None) -> tuple:
    """The HOST side of every translation unit (--cuda-host-only: argument validation, dispatch, plan and workspace-layout
    arithmetic, the stamp ring — no device code) built with AddressSanitizer + UndefinedBehaviorSanitizer into
    build_hostsan/libdcarl_hip_hostsan.so.  It cannot launch anything (the embedded device images are 64 zero bytes); it is what
    tests/test_abi_host_sanitized.py runs the no-GPU tests of the C-ABI against (SURVEY 5 "race detection / sanitizers").
    Returns (library, the sanitizer runtime to LD_PRELOAD into an uninstrumented python)."""
    import glob
    bdir = out_dir or os.path.join(HERE, "build_hostsan")
    os.makedirs(bdir, exist_ok=True)
    san = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined"]
    procs = []
    for src in SOURCES:
        obj = os.path.join(bdir, src.replace(".hip", ".o"))
        cmd = [hipcc(), "--offload-arch=gfx950", "--cuda-host-only", "-O1", "-g", "-std=c++17", "-fPIC", "-fno-honor-nans", *san,
               '-DDCARL_BUILD_ID="hostsan"', "-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    objs = []
    for src, obj, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc (host only, sanitized) failed on {src}:\n{out.decode()}")
        objs.append(obj)
    # every host object refers to its translation unit's device image by a hashed symbol; give each one 64 zero bytes
    nm = subprocess.run(["nm", "-u", *objs], capture_output=True, text=True, check=True).stdout
    syms = sorted({w for line in nm.splitlines() for w in line.split() if w.startswith("__hip_fatbin_")})
    stub_c = os.path.join(bdir, "fatbin_stubs.c")
    with open(stub_c, "w") as f:
        f.write("/* generated by dcarl_amd/build.py build_host_sanitized(): placeholders for the device images of a host-only build */\n")
        for sym in syms:
            f.write(f"const char {sym}[64] __attribute__((aligned(4096))) = {{0}};\n")
    stub_o = stub_c[:-2] + ".o"
    subprocess.check_call(["gcc", "-fPIC", "-c", stub_c, "-o", stub_o])
    lib = os.path.join(bdir, "libdcarl_hip_hostsan.so")
    subprocess.check_call([hipcc(), "-shared", "-fPIC", "-fsanitize=address,undefined", "-shared-libasan", *objs, stub_o, "-ldl", "-o", lib])
    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not rt:
        raise RuntimeError("the AddressSanitizer runtime of ROCm's clang was not found")
    return lib, rt[-1]


if __name__ == "__main__":
    if "--hostsan" in sys.argv:
        print(*build_host_sanitized())
        sys.exit(0)
    vs = ["ab"] if "--only-ab" in sys.argv else [""] + (["ab"] if "--ab" in sys.argv or "--all" in sys.argv else [])
    print(*build_all(force="--force" in sys.argv, verbose=True, variants=tuple(vs)))
