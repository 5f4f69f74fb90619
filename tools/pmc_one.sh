cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for cfg in "prose 1 1" "prose 0 0" "python 1 1" "synth 1 0"; do set -- $cfg
OUT=gpurun_out/pmc_one_$1_$2; rm -rf $OUT; mkdir -p $OUT
CORPUS=$1 EXT=$2 RUNS=$3 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT -o sq -- python tools/one_corpus.py 32768 2>&1 | grep "GB/s"
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc_one_*/*counter_collection.csv')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'tamp_compress' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    m = {k: sum(v)/len(v) for k, v in acc.items()}
    cyc = m['GRBM_GUI_ACTIVE']/8
    print(f.split('/')[1], 'VALU/stream %.0f SALU %.0f LDS %.0f  cycles %.2fM  VALU busy %.0f%%  LDS busy %.0f%%' % (m['SQ_INSTS_VALU']/32768, m['SQ_INSTS_SALU']/32768, m['SQ_INSTS_LDS']/32768, cyc/1e6, 100*m['SQ_ACTIVE_INST_VALU']*4/(1024*cyc), 100*m['SQ_ACTIVE_INST_LDS']/(256*cyc)))
PY
