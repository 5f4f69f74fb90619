cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/c4s; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python tools/config4.py 262144 > $OUT/log 2>&1
grep "GB/s" $OUT/log
python - <<PY
import csv
for r in csv.DictReader(open('$OUT/s_kernel_stats.csv')):
    if 'tamp_dec' in r['Name']: print(r['Name'][:60], r['Calls'], 'avg ns', r['AverageNs'], 'total', r['TotalDurationNs'])
PY
