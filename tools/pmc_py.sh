cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for D in 0 16384 2 1; do
OUT=gpurun_out/pmc_py_$D; rm -rf $OUT; mkdir -p $OUT
TAMP_AMD_DBG=$D WL="glob:/usr/lib/python3.10/*.py" rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT -o p -- python tools/prof_phases.py 8192 > $OUT/log 2>&1
python - <<PY
import csv, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open('$OUT/p_counter_collection.csv')):
    if 'tamp_compress' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
print('dbg=$D ext1 VALU/stream %.0f  (v1 %.0f)' % (acc['SQ_INSTS_VALU'][0]/8192, acc['SQ_INSTS_VALU'][-1]/8192))
PY
done
