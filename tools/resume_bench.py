"""Throughput of the object kernels with everything resident in HBM: N compressor objects fed PIECE bytes per launch
(tamp_batch_compress_resume, op COMPRESS, then FLUSH), and N decoder objects fed the compressed stream in PIECE-byte
pieces with PIECE*4 bytes of output room per launch (tamp_batch_decompress_resume).  The concatenated output of the
compressor objects is checked against the batch kernel's one-shot result, the decoders' against the input.
usage: N=65536 PIECE=128 python tools/resume_bench.py   (needs an MI355X)"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
import tamp_amd
from tamp_amd import _lib, workloads as wl

lib = _lib.load()
n, L, piece = int(os.environ.get('N', 65536)), 4096, int(os.environ.get('PIECE', 128))
dev = torch.device('cuda:0')
rows = wl.synth_text(n, L)
data = torch.from_numpy(rows.reshape(-1)).to(dev)
p = lambda t: C.c_void_p(t.data_ptr())

# ---- reference result: the batch kernel, whole streams ----
off, ln = wl.csr_for_fixed(n, L)
one = tamp_amd.compress_batch(data, torch.from_numpy(off.astype(np.int64)).to(dev), torch.from_numpy(ln.astype(np.int32)).to(dev),
                              max_in_len=L)
torch.cuda.synchronize()

# ---- compressor objects ----
conf = _lib.TampAmdConf(window=10, literal=8, extended=1)
stride = (lib.tamp_amd_encoder_state_size(10) + 15) & ~15
proto = np.zeros(stride, dtype=np.uint8)
assert lib.tamp_amd_encoder_state_init(proto.ctypes.data_as(C.c_void_p), C.byref(conf), 0, 10) == 0
states = torch.from_numpy(np.tile(proto, (n, 1))).to(dev)
cap1 = piece * 2 + 16
out = torch.empty(n * cap1, dtype=torch.uint8, device=dev)
out_off = (torch.arange(n, dtype=torch.int64, device=dev) * cap1)
out_cap = torch.full((n,), cap1, dtype=torch.int32, device=dev)
out_len = torch.empty(n, dtype=torch.int32, device=dev)
status = torch.empty(n, dtype=torch.int8, device=dev)
consumed = torch.empty(n, dtype=torch.int32, device=dev)
in_len = torch.full((n,), piece, dtype=torch.int32, device=dev)
base = torch.arange(n, dtype=torch.int64, device=dev) * L
collected = torch.zeros((n, L + 64), dtype=torch.uint8, device=dev)
fill = torch.zeros(n, dtype=torch.int64, device=dev)
cols = torch.arange(cap1, device=dev)


def gather():
    # append each object's bytes of this launch to its row of `collected` (outside the timed region)
    m = cols[None, :] < out_len[:, None]
    src = out.view(n, cap1)
    dst_col = (fill[:, None] + cols[None, :])
    collected[torch.arange(n, device=dev)[:, None].expand(-1, cap1)[m], dst_col[m]] = src[m]
    fill.add_(out_len.to(torch.int64))


torch.cuda.synchronize()
t_kernel = 0.0
for s in range(L // piece):
    in_off = base + s * piece
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rc = lib.tamp_batch_compress_resume(p(states), stride, 10, _lib.OP_COMPRESS, 0, p(data), p(in_off), p(in_len), p(out),
                                        p(out_off), p(out_cap), p(out_len), p(status), p(consumed), n, _lib.MEM_DEVICE, 0, None)
    torch.cuda.synchronize(); t_kernel += time.perf_counter() - t0
    assert rc == 0 and bool((status == 0).all()) and bool((consumed == piece).all())
    gather()
zero_len = torch.zeros(n, dtype=torch.int32, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
rc = lib.tamp_batch_compress_resume(p(states), stride, 10, _lib.OP_FLUSH, 0, p(data), p(base), p(zero_len), p(out), p(out_off),
                                    p(out_cap), p(out_len), p(status), p(consumed), n, _lib.MEM_DEVICE, 0, None)
torch.cuda.synchronize(); t_kernel += time.perf_counter() - t0
assert rc == 0 and bool((status == 0).all())
gather()
ok = bool((fill.to(torch.int32) == one.out_len).all())
if ok:
    ref = one.out.view(n, -1)[:, : collected.shape[1]] if one.out.numel() >= n * collected.shape[1] else None
    cap_one = one.out.numel() // n
    a = one.out[: n * cap_one].view(n, cap_one)
    w = min(cap_one, collected.shape[1])
    mask = torch.arange(w, device=dev)[None, :] < one.out_len[:, None]
    ok = bool((a[:, :w][mask] == collected[:, :w][mask]).all())
print(f"compressor objects: {n} x {L} B in {piece}-byte pieces, {L//piece}+1 launches: {t_kernel*1e3:8.2f} ms "
      f"{n*L/t_kernel/1e9:6.2f} GB/s in, same bytes as the batch kernel: {ok}", flush=True)

# ---- decoder objects ----
dstride = (lib.tamp_amd_decoder_state_size(10) + 15) & ~15
dproto = np.zeros(dstride, dtype=np.uint8)
assert lib.tamp_amd_decoder_state_init(dproto.ctypes.data_as(C.c_void_p), None, 10) == 0
dstates = torch.from_numpy(np.tile(dproto, (n, 1))).to(dev)
comp = collected  # row i = stream i, fill[i] bytes
comp_flat = comp.reshape(-1)
row = torch.arange(n, dtype=torch.int64, device=dev) * comp.shape[1]
pos = torch.zeros(n, dtype=torch.int64, device=dev)
ocap1 = piece * 4
dout = torch.empty(n * ocap1, dtype=torch.uint8, device=dev)
dout_off = torch.arange(n, dtype=torch.int64, device=dev) * ocap1
dout_cap = torch.full((n,), ocap1, dtype=torch.int32, device=dev)
back = torch.zeros((n, L + ocap1), dtype=torch.uint8, device=dev)
bfill = torch.zeros(n, dtype=torch.int64, device=dev)
dcols = torch.arange(ocap1, device=dev)
t_dec, launches = 0.0, 0
while True:
    take = torch.clamp(fill - pos, max=piece).to(torch.int32)
    in_off = row + pos
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rc = lib.tamp_batch_decompress_resume(p(dstates), dstride, 10, p(comp_flat), p(in_off), p(take), p(dout), p(dout_off),
                                          p(dout_cap), p(out_len), p(status), p(consumed), n, _lib.MEM_DEVICE, 0, None)
    torch.cuda.synchronize(); t_dec += time.perf_counter() - t0
    launches += 1
    assert rc == 0 and bool((status > 0).all())
    m = dcols[None, :] < out_len[:, None]
    back[torch.arange(n, device=dev)[:, None].expand(-1, ocap1)[m], (bfill[:, None] + dcols[None, :])[m]] = dout.view(n, ocap1)[m]
    bfill.add_(out_len.to(torch.int64))
    pos.add_(consumed.to(torch.int64))
    if bool(((pos == fill) & (out_len == 0)).all()) or launches > 400:
        break
ok = bool((bfill == L).all()) and bool((back[:, :L].reshape(-1) == data).all())
print(f"decoder objects:    {n} x {L} B out, {piece}-byte pieces in / {ocap1} B of room, {launches} launches: {t_dec*1e3:8.2f} ms "
      f"{n*L/t_dec/1e9:6.2f} GB/s out, round trip: {ok}", flush=True)
