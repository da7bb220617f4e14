#!/usr/bin/env python3
"""Instruction mix of the main loop of one kernel in a hipcc -S listing (dev tool; no GPU needed).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-gpu-rdc -fno-slp-vectorize -Ispectrogram_inversion_amd/csrc -Iinclude \
          -S --cuda-device-only -o /tmp/k.s <a .hip file that instantiates the kernel>
    python tools/isa_mix.py /tmp/k.s <substring of the mangled kernel name> [loop.s]

Prints the (label, branch) line pairs of every backward branch, then the op-code histogram of the LARGEST loop (the frame loop of
the wave-level kernels) with the count of vector instructions; the optional third argument receives the loop's text.  Static
counts: arms that a steady-state frame does not take (padding, chunk seams) are included.  This is how DESIGN 3.2's s_waitcnt
vmcnt(0) between the output stores and the per-lane frame loop of the n_fft 1024 kernels were found."""
import re,collections,sys
fn=sys.argv[1]; key=sys.argv[2]
lines=open(fn).read().splitlines()
start=[i for i,l in enumerate(lines) if l.startswith('_Z') and key in l.split(':')[0] and ':' in l][0]
end=[i for i,l in enumerate(lines) if i>start and l.strip().startswith('s_endpgm')][0]
body=lines[start:end+1]
labels={}
for i,l in enumerate(body):
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m: labels[m.group(1)]=i
loops=[]
for i,l in enumerate(body):
    m=re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)',l)
    if m:
        t=m.group(1) or m.group(2)
        if t in labels and labels[t]<i: loops.append((labels[t],i))
print(loops)
lo,hi=max(loops,key=lambda p:p[1]-p[0])
cnt=collections.Counter()
for l in body[lo:hi+1]:
    l=l.strip()
    if not l or l.startswith(';') or l.startswith('.') or l.endswith(':'): continue
    op=l.split()[0]
    cnt[op]+=1
tot=sum(cnt.values())
print('total',tot)
valu=sum(v for k,v in cnt.items() if k.startswith('v_'))
print('valu',valu)
for k,v in cnt.most_common(70): print(f'{k:32s}{v}')
if len(sys.argv)>3:
    open(sys.argv[3],'w').write('\n'.join(body[lo:hi+1]))
