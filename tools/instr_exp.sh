cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/instr_exp; rm -rf $OUT; mkdir -p $OUT
for D in ${DLIST:-0 2 1 8 16 32 64}; do
  TAMP_AMD_DBG=$D rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT -o d$D -- python tools/prof_phases.py 8192 > $OUT/d$D.log 2>&1
  python - <<PY
import csv, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open('gpurun_out/instr_exp/d${D}_counter_collection.csv')):
    if 'tamp_compress' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
sh = lambda k: k.replace('SQ_','').replace('INSTS_','I_')
print('dbg=$D', 'ext1:', {sh(k): round(v[0]/8192) for k, v in sorted(acc.items())}, ' v1:', {sh(k): round(v[-1]/8192) for k, v in sorted(acc.items())})
PY
done
