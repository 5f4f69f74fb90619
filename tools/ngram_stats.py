"""Host-side statistics (no device): in-window candidate pairs per position that share a 2-, 3-, 4-, 5-, 6-, 8-gram with the
   query, for the synthetic text and the real-text stand-ins.  Dev tool behind DESIGN.md 3.7.  usage: python tools/ngram_stats.py"""
import os, sys, glob, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
import tamp_amd
from tamp_amd import workloads as wl
W=1024
d = np.frombuffer(bytes(tamp_amd.initialize_dictionary(W)), dtype=np.uint8)
def stats(rows, name):
    tot = {2:0,3:0,4:0,5:0,6:0,8:0}
    maxw = {2:[],3:[],4:[]}
    npos=0
    best_hist = np.zeros(17, dtype=np.int64)
    for r in rows:
        h = np.concatenate([d, r]).astype(np.int64)
        n = len(r)
        # n-gram ids
        for k in tot:
            # key of k bytes
            keys = np.zeros(len(h)-k+1, dtype=object) if k>7 else None
            key = np.zeros(len(h)-k+1, dtype=np.uint64)
            for j in range(min(k,8)):
                key = key*np.uint64(256) + h[j:len(h)-k+1+j].astype(np.uint64)
            # for each query Q=W+q..., count candidates c in [q, Q-1) with key equal
            # do via sorting: group by key, positions sorted
            order = np.argsort(key, kind='stable')
            ks = key[order]
            cnts = np.zeros(len(key), dtype=np.int32)
            # for each group do two-pointer
            start=0
            bounds = np.flatnonzero(np.diff(ks))+1
            groups = np.split(order, bounds)
            for g in groups:
                if len(g)<2: continue
                # g sorted positions
                lo = np.searchsorted(g, g-W, side='left')
                idx = np.arange(len(g))
                cnts[g] = idx-lo
            qc = cnts[W:W+n-k+1]
            tot[k]+= qc.sum()
            if k in maxw:
                m = len(qc)//64*64
                maxw[k].append(np.sort(qc[:m]).reshape(-1,64).max(axis=1).mean())  # sorted-by-length wave max
        npos+=n
    print(name, "pairs/pos:", {k: round(v/npos,2) for k,v in tot.items()}, "sorted wave-max:", {k: round(float(np.mean(v)),2) for k,v in maxw.items()})
rows = wl.synth_text(8,4096)
stats(rows,"synth")
def corpus(patterns, nbytes):
    buf = bytearray()
    for pat in patterns:
        for f in sorted(glob.glob(pat)):
            try: buf += open(f, 'rb').read()
            except Exception: pass
            if len(buf) >= nbytes: return bytes(buf[:nbytes])
    return bytes(buf)
md = corpus(['/opt/skills/guides/*.md','/root/repo/*.md','/usr/share/common-licenses/*','/usr/share/doc/*/copyright'], 1<<22)
py = corpus(['/usr/lib/python3.10/*.py'], 1<<22)
for nm, blob in (("prose",md),("python",py)):
    n=len(blob)//4096
    rows = np.frombuffer(blob[:n*4096],dtype=np.uint8).reshape(n,4096)
    sel = rows[::max(1,n//8)][:8]
    stats(sel, nm)
