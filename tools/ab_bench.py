#!/usr/bin/env python3
"""A/B of library variants on ONE box with bench.py itself: `ab_bench.py "<bench.py arguments>" <variant.so[:ENV=V,...]> ...`
(the shipped library is always included); three interleaved rounds, prints ms per step and the dominant kernel's launch time.
Variants: `python -c "from spectrogram_inversion_amd import build; build.build_lib(extra_flags=['-DX=1'], out='spectrogram_inversion_amd/variants/x.so')"`."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1].split()
libs = [os.path.join(ROOT, "spectrogram_inversion_amd", "libspecinv.so")] + sys.argv[2:]
res = {l: [] for l in libs}
for rnd in range(int(os.environ.get("AB_ROUNDS", "3"))):
    for l in libs:
        path, _, extra = l.partition(":")
        env = dict(os.environ, SPECINV_LIB=os.path.abspath(path))
        for kv in filter(None, extra.split(",")):
            env[kv.split("=")[0]] = kv.split("=")[1]
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-extra", "--no-pmc", "--no-cpu-baseline", "--no-h2d",
                              "--no-check", *args], env=env, capture_output=True, text=True)
        try:
            d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
            res[l].append((d["ms_per_step"], d["roofline"]["launch_ms"]))
        except Exception:
            print(out.stderr[-800:])
for l in libs:
    print(f"{os.path.basename(l):50s} step " + " ".join(f"{v[0]:.3f}" for v in res[l]) + "   launch " + " ".join(f"{v[1]:.4f}" for v in res[l]), flush=True)
