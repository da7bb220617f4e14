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
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/dev.co | grep -E "\.name:|vgpr_count|vgpr_spill|private_segment_fixed|agpr_count|group_segment_fixed" | grep -A5 "$pat" | grep -vE "^--" | tr '\n' ' ' | sed 's/\.name:/\n.name:/g' | sed 's/  */ /g' | cut -c1-260
rm -rf $tmp
