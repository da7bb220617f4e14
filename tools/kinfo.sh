# print register/scratch usage of the fused kernels inside a built library (dev tool)
f=$1
/opt/rocm/lib/llvm/bin/clang-offload-bundler --list --type=o --input=$f 2>/dev/null | head -3
tmp=$(mktemp -d)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$f --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/dev.co 2>/dev/null || \
python3 - "$f" "$tmp/dev.co" <<'PY'
import sys
b=open(sys.argv[1],'rb').read()
i=b.find(b'\x7fELF', b.find(b'__CLANG_OFFLOAD_BUNDLE__'))
# find ELF with machine AMDGPU (0xE0)
pos=0; found=None
while True:
    i=b.find(b'\x7fELF',pos)
    if i<0: break
    if b[i+18:i+20]==b'\xe0\x00': found=i; break
    pos=i+4
open(sys.argv[2],'wb').write(b[found:])
PY
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/dev.co | grep -E "\.name:|vgpr_count|vgpr_spill|private_segment_fixed" | grep -A3 "k_fused" | grep -vE "^--" | paste - - - - | sed 's/  */ /g' | cut -c1-200
rm -rf $tmp
