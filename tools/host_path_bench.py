"""PCIe-inclusive rate of the host-memory boundary (TAMP_AMD_MEM_HOST): BASELINE config 2 through tamp_batch_compress /
tamp_batch_decompress with pageable (numpy) and pinned (tamp_amd_host_alloc) buffers, wall clock around the call.
TAMP_AMD_HOST_CHUNK_MB=100000 makes the whole batch one chunk (no overlap) for comparison.  Dev tool; needs an MI355X."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from tamp_amd import _lib, workloads as wl
from tamp_amd.batch import TampAmdConf

lib = _lib.load()
n, L = int(os.environ.get('N', 65536)), 4096
rows = wl.synth_text(n, L)
cap1 = 4609


def pinned(nbytes, dtype):
    p = lib.tamp_amd_host_alloc(nbytes)
    assert p
    return np.frombuffer((C.c_ubyte * nbytes).from_address(p), dtype=dtype)


def arrays(alloc):
    a = dict(inp=alloc(n * L, np.uint8), in_off=alloc(n * 8, np.uint64), in_len=alloc(n * 4, np.uint32),
             out=alloc(n * cap1, np.uint8), out_off=alloc(n * 8, np.uint64), out_cap=alloc(n * 4, np.uint32),
             out_len=alloc(n * 4, np.uint32), status=alloc(n, np.int8), back=alloc(n * L, np.uint8),
             back_off=alloc(n * 8, np.uint64), back_cap=alloc(n * 4, np.uint32), back_len=alloc(n * 4, np.uint32),
             consumed=alloc(n * 4, np.uint32))
    a['inp'][:] = rows.reshape(-1)
    a['in_off'][:] = np.arange(n, dtype=np.uint64) * L
    a['in_len'][:] = L
    a['out_off'][:] = np.arange(n, dtype=np.uint64) * cap1
    a['out_cap'][:] = cap1
    a['back_off'][:] = np.arange(n, dtype=np.uint64) * L
    a['back_cap'][:] = L
    return a


def ptr(x):
    return C.c_void_p(x.ctypes.data)


def run(kind, alloc):
    a = arrays(alloc)
    conf = TampAmdConf(window=10, literal=8, extended=1)
    best_c, best_d = 1e9, 1e9
    for it in range(4):
        t0 = time.perf_counter()
        rc = lib.tamp_batch_compress(C.byref(conf), None, ptr(a['inp']), ptr(a['in_off']), ptr(a['in_len']), ptr(a['out']),
                                     ptr(a['out_off']), ptr(a['out_cap']), ptr(a['out_len']), ptr(a['status']), n, L,
                                     _lib.MEM_HOST, 0, None)
        t1 = time.perf_counter()
        assert rc == 0 and (a['status'] == 0).all(), (rc, lib.tamp_amd_last_error())
        best_c = min(best_c, t1 - t0)
        t0 = time.perf_counter()
        rc = lib.tamp_batch_decompress(None, 0, 10, ptr(a['out']), ptr(a['out_off']), ptr(a['out_len']), ptr(a['back']),
                                       ptr(a['back_off']), ptr(a['back_cap']), ptr(a['back_len']), ptr(a['status']),
                                       ptr(a['consumed']), n, _lib.MEM_HOST, 0, None)
        t1 = time.perf_counter()
        assert rc == 0, (rc, lib.tamp_amd_last_error())
        best_d = min(best_d, t1 - t0)
    ok = bool((a['back'] == a['inp']).all()) and bool((a['back_len'] == L).all())
    ratio = a['out_len'].sum() / (n * L)
    print(f"{kind:9s} chunk_mb={os.environ.get('TAMP_AMD_HOST_CHUNK_MB', 'default'):8s} compress {best_c*1e3:8.2f} ms "
          f"{n*L/best_c/1e9:6.2f} GB/s in | decompress {best_d*1e3:8.2f} ms {n*L/best_d/1e9:6.2f} GB/s out | "
          f"round trip {'ok' if ok else 'MISMATCH'} ratio {ratio:.3f}", flush=True)


which = sys.argv[1] if len(sys.argv) > 1 else 'both'
if which in ('both', 'pageable'):
    run('pageable', lambda nbytes, dt: np.zeros(nbytes // np.dtype(dt).itemsize, dtype=dt))
if which in ('both', 'pinned'):
    run('pinned', pinned)
