"""Builds libspecinv.so (HIP, gfx950) in-tree with hipcc.

    python -m spectrogram_inversion_amd.build [--force] [-j N]

hipcc cross-compiles without a GPU; the resulting .so sits next to the sources
(spectrogram_inversion_amd/libspecinv.so), is git-ignored and travels to the GPU
box with the repo snapshot.

    SPECINV_BUILD_APPROX=1 python -m spectrogram_inversion_amd.build      (+ the approximate-projection kernels, see with_approx)

The library is two dozen translation units (csrc/*.hip): the plan and the light kernels in
specinv.hip, each family of heavy wave-level kernels in its own tu_*.hip (explicit
instantiations), compiled in parallel and linked once.  Objects and their dependency
files live in csrc/build/<key>/ (git-ignored); only the units whose sources changed
are recompiled.
"""
from __future__ import annotations

import concurrent.futures
import hashlib
import json
import os
import shutil
import subprocess
import sys
import time

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libspecinv.so")
ARCH = "gfx950"
BASE_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
              # the SLP vectoriser's packing choices cost registers in the wave-level FFT kernels: measured on one box,
              # without it k_fused4 0.312 vs 0.318 ms, 2048/256 and 512/128 +10 %, RTISI-LA +14 % (tools/ab_generic.sh)
              "-fno-slp-vectorize"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or add /opt/rocm/bin to PATH)")


def with_approx() -> bool:
    """SPECINV_BUILD_APPROX=1: also build the approximate-projection copies of the float32 wave-level kernels (tu_approx_*.hip - five
    units that compile every kernel header a second time, ~2 CPU-minutes - selected per plan by specinv_plan_set_exact(plan, 0): 3 %
    on the headline step, used by no BASELINE configuration).  Default: not built; tu_noapprox.hip stands in with empty kernel tables."""
    return os.environ.get("SPECINV_BUILD_APPROX", "0") == "1"


def sources():
    names = [f for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    names = [f for f in names if (f != "tu_noapprox.hip" if with_approx() else not f.startswith("tu_approx_"))]
    srcs = [os.path.join(CSRC, f) for f in names]
    hdrs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(PKG_DIR), "include", "specinv.h"))
    return srcs, hdrs


def sources_hash() -> str:
    """sha1 over the kernel sources (csrc/*.hip, csrc/*.h, include/specinv.h): what a stored profile figure was measured on
    (profiles/traffic.json records it; bench.py flags figures taken on other sources as stale)."""
    h = hashlib.sha1()
    srcs, hdrs = sources()
    for p in sorted(srcs + hdrs):
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _units_stamp() -> str:
    return os.path.join(CSRC, "build", "default", "units.txt")


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    srcs, hdrs = sources()
    if any(os.path.getmtime(p) > t for p in srcs + hdrs):
        return True
    try:                                            # (the set of units linked last time: SPECINV_BUILD_APPROX toggled since?)
        with open(_units_stamp()) as fh:
            return fh.read().split() != [os.path.basename(p) for p in srcs]
    except OSError:
        return False


def _deps(depfile: str):
    """Prerequisites listed in a make-style dependency file (-MD)."""
    try:
        with open(depfile) as fh:
            text = fh.read()
    except OSError:
        return None
    text = text.replace("\\\n", " ")
    _, _, rhs = text.partition(":")
    return [p for p in rhs.split() if p]


def _unit_stale(obj: str, dep: str) -> bool:
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    deps = _deps(dep)
    if deps is None:
        return True
    for p in deps:
        try:
            if os.path.getmtime(p) > t:
                return True
        except OSError:
            return True
    return False


def _jobs(n_units: int) -> int:
    if os.environ.get("SPECINV_BUILD_JOBS"):
        return max(1, int(os.environ["SPECINV_BUILD_JOBS"]))
    return max(1, min(n_units, os.cpu_count() or 1))


def build_lib(force: bool = False, verbose: bool = False, extra_flags=(), out: str | None = None,
              jobs: int | None = None) -> str:
    """Compile what is missing or older than its sources, link.  Returns the library path.
    `extra_flags` / `out` build tuning variants next to the default library (their objects get their own directory)."""
    target = out or LIB_PATH
    extra_flags = tuple(extra_flags) + tuple(os.environ.get("SPECINV_EXTRA_FLAGS", "").split())
    if extra_flags and out is None:
        # a variant built over the default library would stay there, "not stale", for every later caller without the flags
        raise ValueError("a build with extra flags (extra_flags= / SPECINV_EXTRA_FLAGS) needs its own out= path")
    if out is None and not force and not is_stale():
        return LIB_PATH
    srcs, _ = sources()
    flags = BASE_FLAGS + list(extra_flags)
    key = hashlib.sha1(" ".join(flags).encode()).hexdigest()[:10] if extra_flags else "default"
    objdir = os.path.join(CSRC, "build", key)
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    units = []
    for s in srcs:
        stem = os.path.splitext(os.path.basename(s))[0]
        units.append((s, os.path.join(objdir, stem + ".o"), os.path.join(objdir, stem + ".d")))
    todo = [u for u in units if force or _unit_stale(u[1], u[2])]

    def compile_one(u):
        src, obj, dep = u
        # (temporaries carry the pid: two processes building at once - ranks, test workers - each finish their own object and
        # the rename is atomic)
        tmp_obj, tmp_dep = f"{obj}.{os.getpid()}.tmp", f"{dep}.{os.getpid()}.tmp"
        cmd = [hipcc, f"--offload-arch={ARCH}", *flags, "-MD", "-MF", tmp_dep, "-MT", obj, "-c", src, "-o", tmp_obj]
        t0 = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        os.replace(tmp_dep, dep)
        os.replace(tmp_obj, obj)
        return os.path.basename(src), time.time() - t0, r.stderr

    if todo:
        if verbose:
            print(f"hipcc --offload-arch={ARCH} {' '.join(flags)}: {len(todo)} of {len(units)} units", flush=True)
        # the longest units first (times of the previous build), so that the tail of the build is short
        times_path = os.path.join(objdir, "times.json")
        try:
            with open(times_path) as fh:
                times = json.load(fh)
        except (OSError, ValueError):
            times = {}
        todo.sort(key=lambda u: -times.get(os.path.basename(u[0]), 1e9))
        with concurrent.futures.ThreadPoolExecutor(jobs or _jobs(len(todo))) as pool:
            for name, dt, err in pool.map(compile_one, todo):
                times[name] = round(dt, 1)
                if verbose:
                    print(f"  {name}: {dt:.0f} s", flush=True)
                if err.strip() and verbose:
                    print(err, flush=True)
        with open(times_path, "w") as fh:
            json.dump(times, fh)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-fno-gpu-rdc", "-Wl,-z,defs",
           *[u[1] for u in units], "-o", f"{target}.{os.getpid()}.tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(f"{target}.{os.getpid()}.tmp", target)
    if key == "default":
        with open(_units_stamp(), "w") as fh:
            fh.write("\n".join(os.path.basename(u[0]) for u in units))
    if key != "default":
        # a tuning variant's objects are not kept (12 MB each): the whole tree, csrc/build included, travels to the GPU box
        import shutil
        shutil.rmtree(objdir, ignore_errors=True)
    return target


if __name__ == "__main__":
    if "--hash" in sys.argv:
        print(sources_hash())
        sys.exit(0)
    j = None
    if "-j" in sys.argv:
        j = int(sys.argv[sys.argv.index("-j") + 1])
    t0 = time.time()
    print(build_lib(force="--force" in sys.argv, verbose=True, jobs=j))
    print(f"{time.time() - t0:.0f} s")
