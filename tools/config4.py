"""BASELINE config 4: decode 1,048,576 pre-compressed 4 KiB streams whose windows are drawn from 2^8..2^12 (interleaved).
   The streams are produced on the device by this library's own compressor (its parity is established elsewhere), laid
   out as one slab, and decoded in one call; a sample of the output is compared with the plain text."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch, tamp_amd
from tamp_amd import workloads as wl
dev = torch.device('cuda:0')
n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20, 4096
wsel = np.arange(n) % 5 + 8
parts = {}
for w in range(8, 13):
    ids = np.nonzero(wsel == w)[0]
    rows = wl.synth_text(len(ids), L, first_index=int(w) * 1000003)
    off, ln = wl.csr_for_fixed(len(ids), L)
    r = tamp_amd.compress_batch(torch.from_numpy(rows.reshape(-1)).to(dev), torch.from_numpy(off.astype(np.int64)).to(dev),
                                torch.from_numpy(ln.astype(np.int32)).to(dev), window=w, max_in_len=L)
    parts[w] = (ids, rows, r)
olen = torch.zeros(n, dtype=torch.int64, device=dev)
for w, (ids, rows, r) in parts.items():
    olen[torch.from_numpy(ids).to(dev)] = r.out_len.to(torch.int64)
in_off = torch.cumsum(olen, 0) - olen
slab = torch.empty(int(olen.sum().item()) + 64, dtype=torch.uint8, device=dev)
for w, (ids, rows, r) in parts.items():
    idt = torch.from_numpy(ids).to(dev)
    lens = r.out_len.to(torch.int64); src_start = r.out_off.to(torch.int64); dst_start = in_off[idt]
    rep = torch.repeat_interleave(torch.arange(len(ids), device=dev), lens)
    within = torch.arange(int(lens.sum().item()), device=dev) - torch.repeat_interleave(torch.cumsum(lens, 0) - lens, lens)
    slab[dst_start[rep] + within] = r.out[src_start[rep] + within]
    del rep, within
torch.cuda.synchronize()
cap = torch.full((n,), L + 8, dtype=torch.int32, device=dev)
for mode in ('wave', 'global', 'split', 'auto'):
    if mode == 'auto': os.environ.pop('TAMP_AMD_DECODER', None)
    else: os.environ['TAMP_AMD_DECODER'] = mode
    ms = []
    for it in range(3):
        d = tamp_amd.decompress_batch(slab, in_off, olen.to(torch.int32), out_cap=cap, timing=True); ms.append(d.kernel_ms)
    ok = bool((d.status == 2).all().item()) and bool((d.out_len == L).all().item())
    for w, (ids, rows, r) in parts.items():
        step = max(1, len(ids) // 2000)
        sel = torch.from_numpy(ids[::step]).to(dev)
        got = torch.stack([d.out[int(o): int(o) + L] for o in d.out_off[sel][:64].tolist()]).cpu().numpy()
        ok = ok and bool((got == rows[::step][:64]).all())
    print(f"config 4 ({n} x 4 KiB, windows 2^8..2^12, {int(olen.sum().item())/2**30:.2f} GiB in) {mode}: {min(ms):8.2f} ms "
          f"{n*L/min(ms)/1e6:6.1f} GB/s out  all status 2 + sample equal: {ok}", flush=True)
