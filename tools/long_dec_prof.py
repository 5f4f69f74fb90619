"""One long v1 stream decoded a few times: the target of `rocprofv3 --kernel-trace --stats` (tools/long_dec_prof.sh).  Dev tool."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import tamp_amd
from tamp_amd import workloads as wl
blob = wl.real_text('prose') + wl.real_text('python') + wl.real_text('markup')
data = (blob * (100_000_000 // len(blob) + 1))[:100_000_000]
c = tamp_amd.compress(data, extended=False)
for rep in range(3):
    r = tamp_amd.decompress_batch([c], out_cap=len(data) + 64, timing=True)
    print(f"kernels {r.kernel_ms:.2f} ms = {len(data)/r.kernel_ms/1e6:.2f} GB/s out", flush=True)
