#!/bin/bash
# Instruction counters of the lane-per-stream decoders on the bench batch, extended and v1 format.  usage: bash tools/dec_pmc2.sh
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
cat > /tmp/dec_one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch, tamp_amd
from tamp_amd import workloads as wl
dev = torch.device('cuda:0')
n, L = 65536, 4096
rows = wl.synth_text(n, L); off, ln = wl.csr_for_fixed(n, L)
data = torch.from_numpy(rows.reshape(-1)).to(dev); off_t = torch.from_numpy(off.astype(np.int64)).to(dev); len_t = torch.from_numpy(ln.astype(np.int32)).to(dev)
r = tamp_amd.compress_batch(data, off_t, len_t, max_in_len=L, extended=bool(int(os.environ.get('EXT','1'))))
ntok = None
for it in range(3):
    d = tamp_amd.decompress_batch(r.out, r.out_off, r.out_len, out_cap=L + 8, timing=True)
print('EXT', os.environ.get('EXT','1'), 'decode ms', d.kernel_ms, 'compressed bytes/stream', float(r.out_len.float().mean()))
PY
for ext in 1 0; do for mode in lane global; do
  OUT=gpurun_out/decpmc2_${ext}_$mode; rm -rf $OUT; mkdir -p $OUT
  EXT=$ext TAMP_AMD_DECODER=$mode rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $OUT -o a -- python /tmp/dec_one.py > $OUT/log 2>&1
  grep "decode ms" $OUT/log
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('$OUT/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'tamp_decompress_kernel' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
m = {k: v[-1] for k, v in acc.items()}
n = 65536
cyc = m['GRBM_GUI_ACTIVE']/8
print('  ext=$ext $mode: per stream VALU %.0f SALU %.0f LDS %.0f VMEM_RD %.0f VMEM_WR %.0f (wave-instr / 64 streams: VALU %.0f) | %.2fM cycles, VALU busy %.0f%%, waves %d' % (m['SQ_INSTS_VALU']/n, m['SQ_INSTS_SALU']/n, m['SQ_INSTS_LDS']/n, m['SQ_INSTS_VMEM_RD']/n, m['SQ_INSTS_VMEM_WR']/n, m['SQ_INSTS_VALU']/n*64, cyc/1e6, 100*m['SQ_ACTIVE_INST_VALU']*4/(1024*cyc), m['SQ_WAVES']))
PY
done; done
