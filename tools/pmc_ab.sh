# usage: bash tools/pmc_ab.sh TAG "corpus ext encoder" ...   -> profiles/ab/<TAG>_<corpus>_<ext>_<encoder>_{sq}.csv + summary line
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
TAG=$1; shift
mkdir -p gpurun_out/ab
for cfg in "$@"; do set -- $cfg
OUT=gpurun_out/ab/${TAG}_$1_$2_$3; rm -rf $OUT; mkdir -p $OUT
CORPUS=$1 EXT=$2 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT -o sq -- python tools/one_corpus.py 32768 2>&1 | grep "GB/s"
CORPUS=$1 EXT=$2 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d $OUT -o sq2 -- python tools/one_corpus.py 32768 2>&1 | grep -c "GB/s" > /dev/null
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/ab/*/')):
    m = {}
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'tamp_compress' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        m.update({k: sum(v)/len(v) for k, v in acc.items()})
    if 'GRBM_GUI_ACTIVE' not in m: continue
    cyc = m['GRBM_GUI_ACTIVE']/8
    print(d.split('/')[2], 'VALU/stream %.0f SALU %.0f LDS %.0f  cycles %.2fM  VALU busy %.0f%%  LDS busy %.0f%% bankconf %.0f%% waitLDS %.2e' % (m['SQ_INSTS_VALU']/32768, m['SQ_INSTS_SALU']/32768, m['SQ_INSTS_LDS']/32768, cyc/1e6, 100*m['SQ_ACTIVE_INST_VALU']*4/(1024*cyc), 100*m['SQ_ACTIVE_INST_LDS']/(256*cyc), 100*m.get('SQ_LDS_BANK_CONFLICT',0)/(256*cyc), m.get('SQ_WAIT_INST_LDS',0)))
PY
