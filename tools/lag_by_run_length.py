"""P(RLE lag | run length) from the reference C on the frozen corpora.  Test-side dev tool: uses oracle/_ref.  usage: lag_by_run_length.py prose|markup|python"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from lag_stats import tokens
from oracle.checker import Ref
from tamp_amd import workloads as wl
name=sys.argv[1]
rows=wl.tile_rows(wl.real_text(name),768)
n=rows.shape[0]; off,ln=wl.csr_for_fixed(n,4096)
res=Ref().compress_batch(rows.reshape(-1),off,ln,window=10,literal=8,extended=True,threads=8)
hist={}  # L -> [runs, lagged]
extlag=0; ext_cnt=0
for i in range(n):
    d=rows[i]; toks=tokens(res.stream(i),4096)
    lagpos=[(p,c,w,k) for (k,p,c,w,wp) in toks if w<c]
    # maximal runs
    j=0
    runs=[]
    while j<4096:
        e=j+1
        while e<4096 and d[e]==d[j]: e+=1
        if e-j>=8: runs.append((j,e))
        j=e
    for (a,b) in runs:
        L=min(b-a,40)
        lag=any(k=='rle' and a<=p<b for (p,c,w,k) in lagpos)
        h=hist.setdefault(L,[0,0]); h[0]+=1; h[1]+=lag
    extlag+=sum(1 for (p,c,w,k) in lagpos if k=='ext')
print(name,'ext lags/chunk',extlag/n)
for L in sorted(hist): print(L,hist[L][0],hist[L][1],'%.2f'%(hist[L][1]/hist[L][0]))
