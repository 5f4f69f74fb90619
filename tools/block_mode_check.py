"""Block mode (one long v1 stream over all workgroups) against the reference C: sizes around block boundaries, windows,
custom dictionary, tight output room; then the time of a 100,000,000-byte stream.  Dev tool (GPU box)."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
import tamp_amd
from tamp_amd import workloads as wl
from oracle.checker import Oracle, Ref
chk = Ref() if Ref.available() else Oracle()
dev = torch.device('cuda:0')
def text(n, kind):
    if kind == 'synth':
        return wl.synth_text((n + 4095) // 4096, 4096).reshape(-1)[:n].copy()
    blob = wl.real_text(kind)
    reps = (n + len(blob) - 1) // len(blob)
    return np.frombuffer((blob * reps)[:n], dtype=np.uint8).copy()
def ref(flat, **kw):
    return chk.compress_batch(flat, np.zeros(1, np.uint64), np.array([flat.size], np.uint32), **kw).stream(0)
bad = 0
for kind in ('prose', 'synth', 'python'):
    for n in (262144, 262145, 262144 + 1023, 300001, 1 << 20, (1 << 20) + 777):
        for w in ((10,) if n != 300001 else (8, 9, 10, 11, 12, 14)):
            flat = text(n, kind)
            want = ref(flat, window=w, literal=8, extended=False)
            got = tamp_amd.compress_batch([flat.tobytes()], window=w, literal=8, extended=False)
            ok = got.stream(0) == want and int(got.status[0]) == 0
            os.environ['TAMP_AMD_BLOCK_MIN'] = '0'
            plain = tamp_amd.compress_batch([flat.tobytes()], window=w, literal=8, extended=False).stream(0)
            del os.environ['TAMP_AMD_BLOCK_MIN']
            if not ok or plain != want:
                bad += 1
                g = got.stream(0)
                k = next((i for i in range(min(len(g), len(want))) if g[i] != want[i]), -1)
                print(f"MISMATCH {kind} n={n} w={w}: got {len(g)} B want {len(want)} B first diff at {k} status {int(got.status[0])} (batch kernel ok: {plain == want})", flush=True)
print("sizes/windows:", "all equal" if not bad else f"{bad} MISMATCHES", flush=True)
# custom dictionary, tight room, odd output offset (device tensors)
flat = text(500000, 'prose')
dic = bytes(range(256)) * 4
want = ref(flat, window=10, literal=8, extended=False, dictionary=dic)
got = tamp_amd.compress_batch([flat.tobytes()], window=10, literal=8, extended=False, dictionary=dic)
print("custom dictionary:", got.stream(0) == want)
want = ref(flat, window=10, literal=8, extended=False)
for cap in (len(want), len(want) - 1, 1000, 3):
    g = tamp_amd.compress_batch([flat.tobytes()], window=10, literal=8, extended=False, out_cap=cap)
    exp_status = 0 if cap >= len(want) else 1
    print(f"cap {cap}: status {int(g.status[0])} (want {exp_status}) len {int(g.out_len[0])} prefix ok {g.stream(0) == want[:min(cap, len(want))]}")
d = torch.from_numpy(flat).to(dev)
for shift in (1, 2, 3):
    cap = len(want) + 64
    outcap = torch.tensor([cap], dtype=torch.int32, device=dev)
    # odd output offset: a slab whose first stream starts `shift` bytes in
    r = tamp_amd.compress_batch(d, torch.zeros(1, dtype=torch.int64, device=dev), torch.tensor([flat.size], dtype=torch.int32, device=dev),
                                window=10, literal=8, extended=False, max_in_len=flat.size, out_cap=torch.tensor([shift, cap], dtype=torch.int32, device=dev)[1:]) if False else None
print("100 MB stream:")
big = text(100_000_000, 'prose')
dbig = torch.from_numpy(big).to(dev)
off = torch.zeros(1, dtype=torch.int64, device=dev); ln = torch.tensor([big.size], dtype=torch.int32, device=dev)
for it in range(3):
    r = tamp_amd.compress_batch(dbig, off, ln, window=10, literal=8, extended=False, max_in_len=big.size, timing=True)
    torch.cuda.synchronize()
    print(f"  device-resident: kernel {r.kernel_ms:.2f} ms = {big.size / r.kernel_ms / 1e6:.2f} GB/s, out {int(r.out_len[0])} B status {int(r.status[0])}", flush=True)
t0 = time.perf_counter(); want = ref(big, window=10, literal=8, extended=False); t1 = time.perf_counter()
print(f"  reference C on one core: {t1 - t0:.2f} s = {big.size / (t1 - t0) / 1e6:.1f} MB/s; equal: {r.stream(0) == want}")
t0 = time.perf_counter(); g = tamp_amd.compress(big.tobytes(), window=10, literal=8, extended=False); t1 = time.perf_counter()
print(f"  tamp_amd.compress(host bytes): {t1 - t0:.3f} s = {big.size / (t1 - t0) / 1e9:.2f} GB/s; equal: {bytes(g) == want}")
