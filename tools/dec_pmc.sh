cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/decpmc; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/dec_one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch, tamp_amd
from tamp_amd import workloads as wl
dev = torch.device('cuda:0')
n, L = 32768, 4096
rows = wl.synth_text(n, L); off, ln = wl.csr_for_fixed(n, L)
data = torch.from_numpy(rows.reshape(-1)).to(dev); off_t = torch.from_numpy(off.astype(np.int64)).to(dev); len_t = torch.from_numpy(ln.astype(np.int32)).to(dev)
r = tamp_amd.compress_batch(data, off_t, len_t, max_in_len=L, extended=bool(int(os.environ.get('EXT','1'))))
cap = torch.full((n,), L, dtype=torch.int32, device=dev)
for it in range(2):
    d = tamp_amd.decompress_batch(r.out, r.out_off, r.out_len, out_cap=cap, timing=True)
print(d.kernel_ms)
PY
run() { TAMP_AMD_DECODER=lane rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT -o $1 -- python /tmp/dec_one.py > $OUT/$1.log 2>&1; }
run a "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
run b "SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY"
python - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob('gpurun_out/decpmc/*_counter_collection.csv')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'tamp_decompress_kernel' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(f.split('/')[-1], {k: round(v[-1]/512) for k, v in acc.items()}, '(per wave of 64 streams, last launch)')
PY
