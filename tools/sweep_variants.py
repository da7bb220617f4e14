#!/usr/bin/env python3
"""Build tuning variants of libspecinv.so (dev tool).  usage: sweep_variants.py name=FLAGS ...
Writes spectrogram_inversion_amd/variants/libspecinv_<name>.so and prints register usage."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectrogram_inversion_amd import build
vd = os.path.join(build.PKG_DIR, "variants")
os.makedirs(vd, exist_ok=True)
procs = []
for spec in sys.argv[1:]:
    name, flags = spec.split("=", 1)
    out = os.path.join(vd, f"libspecinv_{name}.so")
    cmd = [sys.executable, "-c",
           f"import sys; sys.path.insert(0, {os.path.dirname(build.PKG_DIR)!r}); "
           f"from spectrogram_inversion_amd import build; build.build_lib(force=True, extra_flags={flags.split()!r}, out={out!r})"]
    procs.append((name, subprocess.Popen(cmd)))
for name, p in procs:
    p.wait()
    print(name, "rc", p.returncode)
