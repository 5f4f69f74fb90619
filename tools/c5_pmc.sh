cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
WL=telemetry SLEN=256 WINDOW=8 LITERAL=7 TELDICT=1 python tools/prof_phases.py 65536 2>&1 | grep -v amdgpu
OUT=gpurun_out/c5pmc; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT -o sq -- python tools/config5.py 1048576 > $OUT/log 2>&1
tail -3 $OUT/log
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('$OUT/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'tamp_compress' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
m = {k: sum(v)/len(v) for k, v in acc.items()}
n = 1048576
cyc = m['GRBM_GUI_ACTIVE']/8
print('config5 pmc: VALU/msg %.0f SALU %.0f LDS %.0f | cycles %.2fM VALU busy %.0f%% LDS busy %.0f%% waves %.0f' % (m['SQ_INSTS_VALU']/n, m['SQ_INSTS_SALU']/n, m['SQ_INSTS_LDS']/n, cyc/1e6, 100*m['SQ_ACTIVE_INST_VALU']*4/(1024*cyc), 100*m['SQ_ACTIVE_INST_LDS']/(256*cyc), m['SQ_WAVES']))
PY
