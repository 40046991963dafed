import re,sys
# usage: isa_loop.py file.s mangled-substring  -> stats of the innermost loop that holds the first ASMSTART global_load
src=open(sys.argv[1]).read().splitlines()
key=sys.argv[2]
# function range
start=None
for i,l in enumerate(src):
    if l.startswith('_ZN') and key in l and re.match(r'_ZN\S+:', l):
        start=i;break
end=next(i for i in range(start,len(src)) if '.end_amdhsa_kernel' in src[i] or src[i].strip().startswith('s_endpgm'))
fn=src[start:end]
# find first asm load
li=next(i for i,l in enumerate(fn) if 'global_load_dwordx2' in l and '#ASMSTART' in fn[i-1])
# loop header: nearest preceding label with "Inner Loop Header"
hs=max(i for i in range(li) if re.match(r'\.LBB\d+_\d+:',fn[i]) and 'Inner Loop Header' in ''.join(fn[i:i+5]))
lab=fn[hs].split(':')[0]
# back edge: last branch to lab
be=max(i for i,l in enumerate(fn) if re.search(r's_cbranch\w*\s+'+re.escape(lab)+r'\b|s_branch\s+'+re.escape(lab)+r'\b',l))
body=[l.split(';')[0].strip() for l in fn[hs:be+1]]
body=[l for l in body if l and not l.endswith(':') and not l.startswith('.') and not l.startswith(';')]
cls={}
vcc_writer=None
slow=[]
for l in body:
    op=l.split()[0]
    if op.startswith('v_cmp'): c='v_cmp'
    elif op.startswith('v_cndmask_b32_e64'): c='v_cndmask_e64'
    elif op.startswith('v_cndmask'): c='v_cndmask_e32'
    elif op.startswith('v_'):
        c='valu3' if re.search(r'(_e64|bfe|add3|lshl_add|lshl_or|and_or|or3|bitop3|mad_|alignb|perm|mul_|bcnt|mbcnt)',op) else 'valu2'
    elif op.startswith('s_waitcnt') or op.startswith('s_nop'): c='wait'
    elif op.startswith('s_cbranch') or op.startswith('s_branch'): c='branch'
    elif op.startswith('s_'): c='salu'
    elif op.startswith('ds_'): c='lds'
    elif op.startswith('global_') or op.startswith('flat_') or op.startswith('buffer_'): c='vmem'
    else: c='other'
    cls[c]=cls.get(c,0)+1
    # vcc tracking
    if c=='v_cndmask_e32' and vcc_writer and vcc_writer.startswith('s_'):
        slow.append((l,vcc_writer))
    m=re.match(r'(\S+)\s+vcc\b',l)
    if m and not op.startswith('v_cndmask'):
        vcc_writer=op
    if op.startswith('v_cmp') and '_e32' in op: vcc_writer=op
print(lab,'loop instructions:',len(body),cls)
print('v_cndmask_b32_e32 reading an SALU-written VCC:',len(slow))
for s in slow: print('   ',s)
