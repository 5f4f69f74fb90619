#!/bin/bash
# RESOLVE kernel truncated after a phase (libtamp_vars<k>.so built with -DTAMP_SPLIT_STOP=k): time and instruction counts per phase.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
cat > /tmp/dec_spv.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from tamp_amd import _lib
if os.environ.get('TAMP_VAR'): _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libtamp_var%s.so' % os.environ['TAMP_VAR'])
import tamp_amd
from tamp_amd import workloads as wl
dev = torch.device('cuda:0')
n, L = 65536, 4096
rows = wl.synth_text(n, L); off, ln = wl.csr_for_fixed(n, L)
data = torch.from_numpy(rows.reshape(-1)).to(dev); off_t = torch.from_numpy(off.astype(np.int64)).to(dev); len_t = torch.from_numpy(ln.astype(np.int32)).to(dev)
r = tamp_amd.compress_batch(data, off_t, len_t, max_in_len=L)
for it in range(3):
    d = tamp_amd.decompress_batch(r.out, r.out_off, r.out_len, out_cap=L + 8, timing=True)
PY
for v in s1 s2 s3 ""; do
OUT=gpurun_out/spv_$v; rm -rf $OUT; mkdir -p $OUT
TAMP_VAR=$v TAMP_AMD_DECODER=split rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT -o p -- python /tmp/dec_spv.py > $OUT/log 2>&1
python - <<PY
import csv, collections
acc = collections.defaultdict(list); dur = []
for r in csv.DictReader(open('$OUT/p_counter_collection.csv')):
    if 'tamp_decode_resolve' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
for r in csv.DictReader(open('$OUT/p_kernel_trace.csv')):
    if 'tamp_decode_resolve' in r['Kernel_Name']: dur.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
m = {k: v[-1] / 65536 for k, v in acc.items()}
print('stop=[$v] resolve: %.3f ms | per stream VALU %.0f SALU %.0f LDS %.0f cycles %.2fM' % (min(dur) / 1e6, m['SQ_INSTS_VALU'], m['SQ_INSTS_SALU'], m['SQ_INSTS_LDS'], m['GRBM_GUI_ACTIVE'] * 65536 / 8e6))
PY
done
