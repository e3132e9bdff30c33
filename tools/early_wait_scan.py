"""Diagnostic: waits that sit right behind a load in a kernel's ISA (a select on a load's destination, `x = ok ? load(p) : 0`, makes the
compiler wait for the load where the select is written -- a prefetch then never runs ahead).  Poll loops and single dependent loads show
up too; what matters are hits inside batched operand loads.
    hipcc -O3 --offload-arch=gfx950 -std=c++17 -w --cuda-device-only -gline-tables-only -S -o /tmp/ex/abig.s vjf_amd/csrc/vjf_abi.hip
    python tools/early_wait_scan.py [mangled kernel name prefix]        (default: vjf_mega_kernel)
prints: count, (source line of the load, source line of the instruction behind the wait, that instruction)"""
import re,sys
kern=sys.argv[1] if len(sys.argv)>1 else '_Z15vjf_mega_kernel'
src=open('/tmp/ex/abig.s').read().split('\n')
files={}
for l in src:
    m=re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?',l)
    if m: files[int(m.group(1))]=(m.group(3) or m.group(2))
inside=False; cur=None; ins=[]
for l in src:
    if l.startswith(kern): inside=True
    if not inside: continue
    m=re.match(r'\s*\.loc\s+(\d+)\s+(\d+)',l)
    if m: cur=(files.get(int(m.group(1)),'?').split('/')[-1],int(m.group(2))); continue
    t=l.strip()
    if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'): continue
    ins.append((t,cur))
    if t.startswith('s_endpgm'): break
isload=lambda t: re.match(r'(global_load|buffer_load)',t) is not None
hits={}
for i,(t,loc) in enumerate(ins):
    if isload(t):
        for j in range(i+1,min(i+6,len(ins))):
            tj=ins[j][0]
            if isload(tj): break
            if tj.startswith('s_waitcnt') and 'vmcnt(0)' in tj:
                key=(loc, ins[j+1][1] if j+1<len(ins) else None, ins[j+1][0].split()[0] if j+1<len(ins) else '')
                hits[key]=hits.get(key,0)+1
                break
for k,v in sorted(hits.items(), key=lambda x:-x[1])[:50]: print(v,k)
