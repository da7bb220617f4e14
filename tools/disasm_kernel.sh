# disassemble one kernel of a libspecinv build (dev tool): disasm_kernel.sh <lib.so> <mangled-name-substring>
f=$1; pat=$2
tmp=$(mktemp -d)
python3 - "$f" "$tmp/dev.co" <<'PY'
import sys
b=open(sys.argv[1],'rb').read()
pos=0; found=None
while True:
    i=b.find(b'\x7fELF',pos)
    if i<0: break
    if b[i+18:i+20]==b'\xe0\x00': found=i; break
    pos=i+4
open(sys.argv[2],'wb').write(b[found:])
PY
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn $tmp/dev.co | awk -v pat="$pat" '/^[0-9a-f]+ <.*>:/{on=($0 ~ pat)} on{print}'
rm -rf $tmp
