#!/bin/bash
# Same-box comparison of library variants (tamp_amd/libtamp_var<X>.so): kernel time (tools/ab_bench.py) and, per variant,
# one rocprofv3 --pmc pass with the instruction counters.   usage: VARS="0 1 2" bash tools/ab_pmc.sh
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for v in $VARS; do TAMP_VAR=$v python tools/ab_bench.py 2>&1 | grep -v amdgpu.ids; done
for v in $VARS; do
  OUT=gpurun_out/abpmc_$v; rm -rf $OUT; mkdir -p $OUT
  TAMP_VAR=$v AB_ONLY_EXT=${AB_ONLY_EXT:-1} rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT -o sq -- python tools/ab_bench.py > $OUT/log 2>&1
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('$OUT/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'tamp_compress' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
m = {k: sum(v)/len(v) for k, v in acc.items()}
n = 65536
cyc = m['GRBM_GUI_ACTIVE']/8
print('var $v pmc: VALU/stream %.0f SALU %.0f LDS %.0f | cycles %.2fM VALU busy %.0f%% LDS busy %.0f%% bank-conflict cycles/stream %.0f' % (m['SQ_INSTS_VALU']/n, m['SQ_INSTS_SALU']/n, m['SQ_INSTS_LDS']/n, cyc/1e6, 100*m['SQ_ACTIVE_INST_VALU']*4/(1024*cyc), 100*m['SQ_ACTIVE_INST_LDS']/(256*cyc), m['SQ_LDS_BANK_CONFLICT']/n))
PY
done
