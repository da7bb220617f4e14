"""Register budgets of the shipped kernels, read from the code object inside libspecinv.so (no GPU needed).

The wave-level kernels are tuned to a waves-per-SIMD target through `__launch_bounds__`; a kernel that silently loses its
bounds (e.g. a declaration instantiated ahead of the definition, as happened to k_rtisi_fast when the library was split into
translation units: 128 registers, 269 spilled, C3 2.6 x slower) still passes every parity test.  This test pins the budgets of
the kernels the BASELINE configurations run."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "spectrogram_inversion_amd", "libspecinv.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def _kernel_table():
    """{demangled-ish kernel name: (vgprs, spilled vgprs, agprs)} over every gfx950 code object bundled in the library."""
    blob = open(LIB, "rb").read()
    out, pos = {}, 0
    tmp = os.path.join(ROOT, "spectrogram_inversion_amd", "csrc", "build", "_code_object.tmp")
    os.makedirs(os.path.dirname(tmp), exist_ok=True)
    while True:
        i = blob.find(b"\x7fELF", pos)
        if i < 0:
            break
        pos = i + 4
        if blob[i + 18:i + 20] != b"\xe0\x00":          # e_machine: EM_AMDGPU
            continue
        nxt = blob.find(b"\x7fELF", pos)
        with open(tmp, "wb") as fh:
            fh.write(blob[i:nxt if nxt > 0 else len(blob)])
        notes = subprocess.run([READELF, "--notes", tmp], capture_output=True, text=True).stdout
        for m in re.finditer(r"\.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.vgpr_count:\s+(\d+)\s+\.vgpr_spill_count:\s+(\d+)", notes, re.S):
            out[m.group(2)] = (int(m.group(3)), int(m.group(4)), int(m.group(1)))
    if os.path.exists(tmp):
        os.remove(tmp)
    return out


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(LIB) or not os.path.exists(READELF):
        pytest.skip("libspecinv.so / llvm-readelf not available")
    t = _kernel_table()
    assert len(t) > 150, len(t)
    return t


def _find(table, *parts, approx=False):
    """the kernel whose mangled name holds every part, in namespace fast (default: the reference's operation order in the projection)
    or fast_approx (the copy with the approximate projection)"""
    hits = [k for k in table if all(p in k for p in parts) and ("11fast_approx" in k) == approx]
    assert len(hits) == 1, (parts, hits)
    return table[hits[0]]


def test_headline_kernels_keep_their_register_budgets(kernels):
    # C2: k_fused4_td<16, early, eval> - two waves per SIMD (<= 256 registers); the plain late launch (86 of a step's 100) spills
    # nothing, the early one (14) a couple; the fused evaluating variants are not launched at this shape (k_eval_td)
    for early, ev, max_spill in (("Lb0E", "Lb0E", 0), ("Lb1E", "Lb0E", 4), ("Lb0E", "Lb1E", 64), ("Lb1E", "Lb1E", 64)):
        v, sp, _ = _find(kernels, "11k_fused4_tdILi16E" + early + ev)
        assert v <= 256 and sp <= max_spill, (early, ev, v, sp)
    # C4: k_fused4<8, ADMM> - three waves per SIMD (<= 168 registers, a handful spilled)
    v, sp, _ = _find(kernels, "8k_fused4ILi8ELi1ELb0E")
    assert v <= 168 and sp <= 8, (v, sp)
    # C3: k_rtisi_fast<16, 256, 4> - one wave per SIMD, the whole 512-entry file, no spills
    v, sp, a = _find(kernels, "12k_rtisi_fastILi16ELi256ELi4E")
    assert v > 256 and sp == 0, (v, sp, a)
    # the approximate-projection copy of the headline kernel (SPECINV_BUILD_APPROX=1 builds only) keeps the two-waves budget as well
    if any("11fast_approx" in k for k in kernels):
        v, sp, _ = _find(kernels, "11k_fused4_tdILi16ELb0ELb0E", approx=True)
        assert v <= 256 and sp == 0, (v, sp)
    # C5: k_objective_logmel<16, 9, false, true> (the mel filterbank as bands) and <16, 5> (the same on the matrix cores) - two
    # waves per SIMD (<= 256 registers), no spills
    for name in ("18k_objective_logmelILi16ELi9ELb0ELb1E", "18k_objective_logmelILi16ELi5ELb0ELb0E"):
        v, sp, _ = _find(kernels, name)
        assert v <= 256 and sp == 0, (name, v, sp)


def test_no_shipped_wave_level_kernel_spills_heavily(kernels):
    """k_objective_logmel<8, 8> compiled to 256 registers + 978 spilled and ran 1.8 x slower than its neighbours <8, 7> / <8, 9> for
    a whole round before anybody looked: no kernel of the wave-level family may spill more than a hundred registers (the known
    heavy ones: the early evaluating launch, k_rtisi_fast with 512-sample hops at look-ahead 8)."""
    bad = {k: v for k, v in kernels.items() if ("4fast" in k or "11fast_approx" in k) and v[1] > 140}
    assert not bad, bad


def test_no_wave_level_kernel_is_capped_at_the_default_bounds(kernels):
    """A wave-level kernel compiled without its launch bounds gets the 1024-thread default: exactly 128 registers plus a large
    spill.  None of them may look like that."""
    bad = {k: v for k, v in kernels.items() if "4fast" in k and v[0] == 128 and v[1] > 40}
    assert not bad, bad
