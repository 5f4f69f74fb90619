#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of library variants.  usage: VARS="0 5" bash tools/ab_traffic.sh
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for v in $VARS; do
  for c in FETCH_SIZE WRITE_SIZE; do
    OUT=gpurun_out/abtr_${v}_$c; rm -rf $OUT; mkdir -p $OUT
    TAMP_VAR=$v AB_ONLY_EXT=1 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT -o p -- python tools/ab_bench.py > $OUT/log 2>&1
  done
  python - <<PY
import csv, glob
def mean(c):
    vals = [float(r['Counter_Value']) for f in glob.glob('gpurun_out/abtr_${v}_%s/*counter_collection.csv' % c) for r in csv.DictReader(open(f)) if 'tamp_compress' in r['Kernel_Name']]
    return sum(vals)/len(vals)
f, w = mean('FETCH_SIZE'), mean('WRITE_SIZE')
print('var $v traffic: FETCH %.0f KB x2 = %.1f MB, WRITE %.1f MB, total %.1f MB = %.2f x algorithmic (416.4 MB)' % (f, 2*f*1024/1e6, w*1024/1e6, (2*f+w)*1024/1e6, (2*f+w)*1024/416387342))
PY
done
