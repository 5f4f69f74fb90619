cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
cat > /tmp/dec_sp.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch, tamp_amd
from tamp_amd import workloads as wl
dev = torch.device('cuda:0')
n, L = int(os.environ.get('N', '65536')), 4096
rows = wl.synth_text(n, L); off, ln = wl.csr_for_fixed(n, L)
data = torch.from_numpy(rows.reshape(-1)).to(dev); off_t = torch.from_numpy(off.astype(np.int64)).to(dev); len_t = torch.from_numpy(ln.astype(np.int32)).to(dev)
r = tamp_amd.compress_batch(data, off_t, len_t, max_in_len=L, extended=bool(int(os.environ.get('EXT','1'))))
for it in range(3):
    d = tamp_amd.decompress_batch(r.out, r.out_off, r.out_len, out_cap=L + 8, timing=True)
print('decode ms', d.kernel_ms, bool((d.status==2).all().item()))
PY
OUT=gpurun_out/sp; rm -rf $OUT; mkdir -p $OUT
TAMP_AMD_DECODER=split rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python /tmp/dec_sp.py > $OUT/log 2>&1
grep "decode ms" $OUT/log
python - <<PY
import csv
for r in csv.DictReader(open('$OUT/s_kernel_stats.csv')):
    if 'tamp' in r['Name']: print(r['Name'][:60], r['Calls'], 'avg ns', r['AverageNs'])
PY
TAMP_AMD_DECODER=split rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $OUT -o p -- python /tmp/dec_sp.py > $OUT/log2 2>&1
python - <<PY
import csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open('$OUT/p_counter_collection.csv')):
    if 'tamp_decode' in r['Kernel_Name']: acc[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
n = int('${N:-65536}')
for k, m in acc.items():
    m = {c: v[-1] for c, v in m.items()}
    cyc = m['GRBM_GUI_ACTIVE']/8
    print(k, 'per stream VALU %.0f SALU %.0f LDS %.0f VMEM_RD %.0f VMEM_WR %.0f | %.2fM cycles VALU busy %.0f%% waves %d' % (m['SQ_INSTS_VALU']/n, m['SQ_INSTS_SALU']/n, m['SQ_INSTS_LDS']/n, m['SQ_INSTS_VMEM_RD']/n, m['SQ_INSTS_VMEM_WR']/n, cyc/1e6, 100*m['SQ_ACTIVE_INST_VALU']*4/(1024*cyc), m['SQ_WAVES']))
PY
