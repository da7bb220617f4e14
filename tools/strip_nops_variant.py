#!/usr/bin/env python3
"""Experiment (upper bound, not a build step): how much do the `s_nop 0` the compiler pads the frame loop of k_fused4_td<16> with
cost?  Rebuilds ONE translation unit through hipcc's own pipeline (`-save-temps -###`), deletes every `s_nop 0` inside the named
kernels of the device assembly before it is assembled, and links a variant library from the default build's other objects.
The padding guards a forwarding hazard the compiler assumes after inline-asm blocks and between dependent packed instructions
(DESIGN 3.2 (9)); if the variant's results stay bit-identical the hazard is not real for these instructions, and the timing
difference is what restructuring the sources to avoid the padding could win at most.

    python tools/strip_nops_variant.py            # -> spectrogram_inversion_amd/variants/libspecinv_nonop.so
"""
import os
import shlex
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "spectrogram_inversion_amd", "csrc")
WORK = os.path.join(ROOT, "gpurun_out", "nonop_build")
UNIT = sys.argv[1] if len(sys.argv) > 1 else "tu_td4"
KERNELS = ["_ZN7specinv4fast11k_fused4_tdILi16ELb0ELb0EEEvNS0_8FastArgsE", "_ZN7specinv4fast11k_fused4_tdILi16ELb1ELb0EEEvNS0_8FastArgsE"]

shutil.rmtree(WORK, ignore_errors=True)
os.makedirs(WORK)
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-fno-slp-vectorize",
         f"-I{CSRC}", f"-I{os.path.join(ROOT, 'include')}"]
r = subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-c", os.path.join(CSRC, UNIT + ".hip"), "-o", UNIT + ".o", "-save-temps", "-###"],
                   cwd=WORK, capture_output=True, text=True)
cmds = [l.strip() for l in r.stderr.splitlines() if l.startswith(' "')]
assert len(cmds) >= 10, r.stderr[-2000:]
dev_s = os.path.join(WORK, f"{UNIT}-hip-amdgcn-amd-amdhsa-gfx950.s")
removed = 0
for c in cmds:
    args = shlex.split(c)
    if "-cc1as" in args and "amdgcn-amd-amdhsa" in args:      # the device assembler is next: patch its input
        # only the padding after an inline-asm block or a packed instruction (the assumed forwarding hazard): the `s_nop` that
        # follow loads, SGPR writes or compares guard real hazards (a blanket removal faults)
        out, inside, prev = [], False, ""
        for line in open(dev_s).read().split("\n"):
            if any(line.startswith(k + ":") for k in KERNELS):
                inside = True
            if inside and "s_endpgm" in line:
                inside = False
            t = line.strip()
            if inside and t == "s_nop 0" and (prev == ";;#ASMEND" or prev.startswith("v_pk_")):
                removed += 1
                continue
            if t and not t.startswith(";") or t.startswith(";;#ASM"):
                prev = t
            out.append(line)
        open(dev_s, "w").write("\n".join(out))
    p = subprocess.run(args, cwd=WORK, capture_output=True, text=True)
    if p.returncode:
        raise SystemExit(f"{args[0]} failed:\n{p.stderr[-2000:]}")
print(f"removed {removed} s_nop 0 from {len(KERNELS)} kernels of {UNIT}")
objdir = os.path.join(CSRC, "build", "default")
objs = [os.path.join(objdir, f) for f in sorted(os.listdir(objdir)) if f.endswith(".o") and f != UNIT + ".o"]
objs.append(os.path.join(WORK, UNIT + ".o"))
target = os.path.join(ROOT, "spectrogram_inversion_amd", "variants", "libspecinv_nonop.so")
os.makedirs(os.path.dirname(target), exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-Wl,-z,defs", *objs, "-o", target], check=True)
print(target)
