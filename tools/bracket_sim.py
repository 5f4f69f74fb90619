"""Host-side simulation of the compress kernel's bucket brackets (no device): per query the entries its bracket holds
   (tile-ordered bigram buckets), the exact bigram pairs among them, and the lock-step cost of scanning them
   64 queries at a time after the sort by bracket length -- for 256- / 64-position tiles and 2048 / 4096 buckets.  Dev tool
   behind DESIGN.md 3.7.  usage: python tools/bracket_sim.py"""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import tamp_amd
from tamp_amd import workloads as wl
W=1024; BLK=1536
d = np.frombuffer(bytes(tamp_amd.initialize_dictionary(W)), dtype=np.uint8)
def analyse(rows, name, tile=256, hbits=11):
    tot_useful=0; tot_lock=0; tot_lock_split=0; nq=0; tot_exact=0
    for r in rows:
        for e0 in (0,1536):
            # epoch buffer: window (last W bytes before e0) + block
            hist = np.concatenate([d, r]).astype(np.uint32)
            buf = hist[e0:e0+W+BLK+1]
            NE = W+BLK
            big = (buf[:NE] | (buf[1:NE+1]<<8))
            h = ((big*40503)&0xFFFF)>>(16-hbits)
            pos = np.arange(NE)
            # bracket for query q (0..BLK): entries in same bucket with tile(q) <= tile(pos) <= tile(q+W)
            L = np.zeros(BLK, dtype=np.int64); X = np.zeros(BLK, dtype=np.int64)
            order = np.argsort(h, kind='stable')
            hs = h[order]; ps = pos[order]
            starts = np.searchsorted(hs, np.arange(1<<hbits)); ends = np.searchsorted(hs, np.arange(1<<hbits), side='right')
            for q in range(BLK):
                b = h[W+q]
                pp = ps[starts[b]:ends[b]]
                lo = (q//tile)*tile; hi = ((W+q)//tile+1)*tile
                L[q] = np.count_nonzero((pp>=lo)&(pp<hi))
                bb = big[pp]
                X[q] = np.count_nonzero((pp>=q)&(pp<=W+q-2)&(bb==big[W+q]))
            s = np.sort(L)[::-1]
            g = s.reshape(-1,64)
            tot_lock += g.max(axis=1).sum()*64
            tot_useful += L.sum(); tot_exact += X.sum(); nq+=BLK
            # splitting: lanes-per-query k chosen per wave so that max/k... simulate: items of <=8 entries
            items = []
            for l in L:
                k = max(1,(l+7)//8)
                items += [ (l+k-1)//k ]*k
            it = np.sort(np.array(items))[::-1]
            pad = (-len(it))%64
            it = np.concatenate([it, np.zeros(pad,dtype=it.dtype)]).reshape(-1,64)
            tot_lock_split += it.max(axis=1).sum()*64
    print(f"{name} tile={tile} hbits={hbits}: exact pairs/q {tot_exact/nq:.2f} scanned/q {tot_useful/nq:.2f} lockstep lane-iters/q {tot_lock/nq:.2f} (eff {tot_useful/tot_lock:.2f}) split<=8: {tot_lock_split/nq:.2f}")
rows = wl.synth_text(6,4096)
analyse(rows,'synth')
analyse(rows,'synth',tile=64)
analyse(rows,'synth',hbits=12)
analyse(rows,'synth',tile=64,hbits=12)
