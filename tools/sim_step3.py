"""CPU model (test tooling): depth of the automaton state right after every CJK lead byte of the cfg 3 text -- how often a
three-byte step from the root row could apply.  Run from the repository root."""
import sys
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np
from aha_amd import synth
blob, offs, nf = synth.keys(3)
corpus, doc = synth.corpus(3, blob, offs, nf, n_bytes=1<<20, doc_bytes=1<<18)
keys=[bytes(blob[int(offs[i]):int(offs[i+1])]) for i in range(len(offs)-1)]
# trie
children=[{}]; depth=[0]
for k in keys:
    s=0
    for c in k:
        n=children[s].get(c)
        if n is None:
            n=len(children); children.append({}); depth.append(depth[s]+1); children[s][c]=n
        s=n
# fail links BFS
from collections import deque
fail=[0]*len(children)
q=deque()
for c,n in children[0].items(): q.append(n)
while q:
    s=q.popleft()
    for c,n in children[s].items():
        f=fail[s]
        while f and c not in children[f]: f=fail[f]
        fail[n]=children[f].get(c,0) if children[f].get(c,0)!=n else 0
        q.append(n)
text=corpus.tobytes()
s=0; land1=0; land1_cjk=0; cjk=0; n=len(text)
hist={}
for i,c in enumerate(text):
    if c==0: s=0; continue
    prev=s
    while s and c not in children[s]: s=fail[s]
    s=children[s].get(c,0)
    if 0xE4<=c<=0xE9:
        cjk+=1
        hist[depth[s]]=hist.get(depth[s],0)+1
        if depth[s]==1:
            land1_cjk+=1
print("bytes",n,"cjk lead bytes",cjk, cjk/n, "state depth after lead byte:", sorted(hist.items()))
