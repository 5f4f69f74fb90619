"""BASELINE config 5: 256-byte 7-bit telemetry messages, shared custom dictionary, window=8 literal=7 extended=1
   (header 0x16).  Default: one GPU's share at 8 GPUs (16 M / 8 = 2,097,152 messages); `python tools/config5.py 16777216`
   runs all 16 M on one device.  A few messages carry a byte >= 0x80 and must come back TAMP_EXCESS_BITS (-2); a sample
   is compared with the oracle; everything is decoded again and compared."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch, tamp_amd
from tamp_amd import workloads as wl
from oracle.checker import Oracle
dev = torch.device('cuda:0')
n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 21, 256
d = wl.telemetry_dictionary(bytes(tamp_amd.initialize_dictionary(256, literal=7)))
CH = 1 << 21
data = torch.empty(n * L, dtype=torch.uint8, device=dev)
bad_ids = [5, n // 3, n - 2]
sample_rows = None
for c0 in range(0, n, CH):                      # generate on the host in pieces, keep only the device copy
    k = min(CH, n - c0)
    rows = wl.telemetry(k, L, first_index=c0).copy()
    for b in bad_ids:
        if c0 <= b < c0 + k: rows[b - c0, 40] = 0xC3
    if c0 == 0: sample_rows = rows[:4096].copy()
    data[c0 * L:(c0 + k) * L] = torch.from_numpy(rows.reshape(-1)).to(dev)
    del rows
off_t = torch.arange(n, dtype=torch.int64, device=dev) * L
len_t = torch.full((n,), L, dtype=torch.int32, device=dev)
ms = []
for it in range(3):
    r = tamp_amd.compress_batch(data, off_t, len_t, window=8, literal=7, dictionary=d, max_in_len=L, timing=True)
    ms.append(r.kernel_ms)
st = r.status.cpu().numpy()
bad = np.nonzero(st != 0)[0].tolist()
ok_status = bad == sorted(bad_ids) and all(int(st[b]) == -2 for b in bad_ids)
off, ln = wl.csr_for_fixed(4096, L)
want = Oracle().compress_batch(sample_rows.reshape(-1), off, ln, window=8, literal=7, dictionary=d, threads=8)
ok_sample = all(r.stream(i) == want.stream(i) and int(st[i]) == int(want.status[i]) for i in range(4096))
olen = r.out_len.to(torch.int64)
print(f"config 5 compress: {n} x {L} B, {n*L/2**30:.2f} GiB in, header 0x{int(r.out[0]):02x}: {min(ms):8.2f} ms {n*L/min(ms)/1e6:6.1f} GB/s in, "
      f"ratio {float(olen.sum().item())/(n*L):.3f}; EXCESS_BITS exactly at {bad_ids}: {ok_status}; first 4096 == oracle: {ok_sample}", flush=True)
cap = torch.full((n,), L + 8, dtype=torch.int32, device=dev)
dms = []
for it in range(3):
    back = tamp_amd.decompress_batch(r.out, r.out_off, r.out_len, out_cap=cap, dictionary=d, timing=True); dms.append(back.kernel_ms)
good = torch.ones(n, dtype=torch.bool, device=dev); good[torch.tensor(bad_ids, device=dev)] = False
got = back.out.view(-1)[: n * (L + 8)].view(n, L + 8)[:, :L] if back.out.numel() >= n * (L + 8) else None
eq = bool((got[good] == data.view(n, L)[good]).all().item()) if got is not None else None
print(f"config 5 decode  : {min(dms):8.2f} ms {n*L/min(dms)/1e6:6.1f} GB/s out; status 2 for all good streams: "
      f"{bool((back.status[good] == 2).all().item())}; round trip equal: {eq}")
