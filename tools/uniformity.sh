#!/bin/bash
# dev: what LLVM's uniformity analysis thinks of a kernel of rc_correct.hip -- the root sources of
# divergence (with source lines) and the number of divergent branches.  A wave-uniform search loop that
# shows up here as divergent runs on vector registers and EXEC-mask branches.
# usage: tools/uniformity.sh _Z9k_correctILi192ELb0ELb0EEv14rc_kernel_args   [EXTRA='-DRC_K3_WAVES=6']
cd "$(dirname "$0")/../rcorrector_amd/csrc"
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -gline-tables-only -S -emit-llvm --cuda-device-only $EXTRA rc_correct.hip -o /tmp/rc.ll 2>/dev/null
/opt/rocm/lib/llvm/bin/opt -passes='print<uniformity>' -disable-output /tmp/rc.ll 2> /tmp/unif.txt
python3 - "$1" <<'PY'
import re,collections,sys
fn=sys.argv[1]
u=open('/tmp/unif.txt').read()
i=u.index("UniformityInfo for function '%s'"%fn)
j=u.find("UniformityInfo for function",i+10)
u=u[i:j if j>0 else None]
open('/tmp/unif_fn.txt','w').write(u)
ll=open('/tmp/rc.ll').read()
loc={}
for m in re.finditer(r'^(!\d+) = !DILocation\(line: (\d+), column: (\d+), scope: (!\d+)(?:, inlinedAt: (!\d+))?\)',ll,re.M):
    loc[m.group(1)]=(int(m.group(2)),int(m.group(3)),m.group(4),m.group(5))
defs={};div=set()
for l in u.split('\n'):
    m=re.match(r'\s*(DIVERGENT:)?\s*(%\d+) = (.*)',l)
    if m and not l.startswith(('Value','Used')):
        defs[m.group(2)]=m.group(3)
        if m.group(1): div.add(m.group(2))
nbr=len(re.findall(r'DIVERGENT:\s+br i1',u))
print("divergent values",len(div),"divergent branches",nbr)
for v in sorted(div,key=lambda x:int(x[1:])):
    d=defs[v]
    if d.startswith('phi'): continue
    ops=[o for o in re.findall(r'%\d+',d) if o in div and o!=v]
    if not ops:
        m=re.search(r'!dbg (!\d+)',d); ch=[]
        dd=m.group(1) if m else None
        while dd and dd in loc:
            ch.append(loc[dd][:2]); dd=loc[dd][3]
        print(v,d[:90],ch)
PY
