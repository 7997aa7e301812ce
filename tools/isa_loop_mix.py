#!/usr/bin/env python3
"""instruction mix of the hottest loop of a kernel from `hipcc -S` output, common path only (code that sits inline behind an `s_cbranch_vccz` to a later label of the loop -- a rare path -- is counted apart).
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I 3d-re-gen_amd/csrc -S --cuda-device-only -o attn.s 3d-re-gen_amd/csrc/attn.hip
    python tools/isa_loop_mix.py attn.s <mangled kernel name> [<name>=<file to dump the loop to>] ..."""
import re, collections, sys
src=open(sys.argv[1]).read().split('\n')
def analyse(sym, dump=None):
    i=next(k for k,l in enumerate(src) if l.startswith(sym+':'))
    j=i
    while not src[j].startswith('.Lfunc_end'): j+=1
    b=src[i:j]
    labels={l.split(':')[0]:k for k,l in enumerate(b) if re.match(r'^\.LBB\d+_\d+:',l)}
    loops=[]
    for k,l in enumerate(b):
        m=re.match(r'\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)',l)
        if m and m.group(1) in labels and labels[m.group(1)]<k: loops.append((labels[m.group(1)],k))
    s,e=max(loops,key=lambda se: sum(1 for l in b[se[0]:se[1]] if 'v_mfma' in l))
    common=collections.Counter(); rare=collections.Counter(); skip_until=None; lines=[]
    for k in range(s,e+1):
        l=b[k]
        m=re.match(r'^(\.LBB\d+_\d+):',l)
        if m:
            if skip_until==m.group(1): skip_until=None
            continue
        if not l.startswith('\t') or l.strip().startswith(';') or l.startswith('\t.'): continue
        ins=l.strip().split(';')[0].strip(); op=ins.split()[0]
        if skip_until: rare[op]+=1; continue
        common[op]+=1; lines.append(ins)
        m=re.match(r's_cbranch_vccz\s+(\.LBB\d+_\d+)',ins)
        if m and labels[m.group(1)]>k and labels[m.group(1)]<=e: skip_until=m.group(1)
    nm=common['v_mfma_f32_32x32x16_bf16']
    valu=sum(v for k,v in common.items() if k.startswith('v_') and not k.startswith('v_mfma'))
    print(sym[28:80],'COMMON path: instrs',sum(common.values()),'mfma',nm,'valu',valu,'valu/mfma %.2f'%(valu/max(nm,1)), '| rare instrs',sum(rare.values()))
    print('  ',sorted(common.items(), key=lambda kv:-kv[1]))
    if dump: open(dump,'w').write('\n'.join(lines))
for a in sys.argv[2:]:
    if '=' in a: sym,d=a.split('='); analyse(sym,d)
    else: analyse(a)
