cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/spv; rm -rf $OUT; mkdir -p $OUT
python - <<'PY'
import re
s = open('/tmp/dec_sp.py').read() if False else None
PY
sed -e "s#import numpy as np, torch, tamp_amd#import numpy as np, torch\nfrom tamp_amd import _lib\nif os.environ.get('TAMP_VAR'): _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libtamp_var%s.so' % os.environ['TAMP_VAR'])\nimport tamp_amd#" /tmp/dec_sp.py > /tmp/dec_spv.py
for v in "" j; do
TAMP_VAR=$v TAMP_AMD_DECODER=split rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s$v -- python /tmp/dec_spv.py > $OUT/log$v 2>&1
python - <<PY
import csv
for r in csv.DictReader(open('$OUT/s${v}_kernel_stats.csv')):
    if 'tamp_decode' in r['Name']: print('var[$v]', r['Name'][:50], 'avg ns', r['AverageNs'])
PY
done
