"""Builds libspecinv.so (HIP, gfx950) in-tree with hipcc.

    python -m spectrogram_inversion_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so sits next to the sources
(spectrogram_inversion_amd/libspecinv.so), is git-ignored and travels to the GPU
box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libspecinv.so")
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or add /opt/rocm/bin to PATH)")


def sources():
    hdrs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(PKG_DIR), "include", "specinv.h"))
    return [os.path.join(CSRC, "specinv.hip")], hdrs


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    srcs, hdrs = sources()
    return any(os.path.getmtime(p) > t for p in srcs + hdrs)


def build_lib(force: bool = False, verbose: bool = False, extra_flags=(), out: str | None = None) -> str:
    """Compile if missing or older than any source.  Returns the library path.
    `extra_flags` / `out` build tuning variants next to the default library."""
    global_out = out or LIB_PATH
    if out is None and not force and not is_stale():
        return LIB_PATH
    srcs, _ = sources()
    extra_flags = tuple(extra_flags) + tuple(os.environ.get("SPECINV_EXTRA_FLAGS", "").split())
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
           # the SLP vectoriser's packing choices cost registers in the wave-level FFT kernels: measured on one box,
           # without it k_fused4 0.312 vs 0.318 ms, 2048/256 and 512/128 +10 %, RTISI-LA +14 % (tools/ab_generic.sh)
           "-fno-slp-vectorize",
           *extra_flags, *srcs, "-o", global_out + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(global_out + ".tmp", global_out)
    return global_out


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
