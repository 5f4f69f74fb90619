# VALU wave-instructions per stream of each section of the tile kernel: the -DTAMP_TILE_DBG build runs a section twice
# when its bit of TAMP_AMD_DBG is set (1 scan loop, 2 whole match loop, 4 index, 8 fill, 16 query sort, 32 jump tables).
# usage (GPU box): CORPUS=synth EXT=1 bash tools/tile_sections.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export TAMP_AMD_LIB=$GRAFT_REPO_ROOT/tamp_amd/libtamp_amd_tdbg.so TAMP_AMD_ENCODER=tile
N=${N:-16384}
for dbg in 0 1 2 4 8 16 32; do
OUT=gpurun_out/tsec/${CORPUS:-synth}_${EXT:-1}_$dbg; rm -rf $OUT; mkdir -p $OUT
TAMP_AMD_DBG=$dbg rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT -o sq -- python tools/one_corpus.py $N 2>&1 | grep -c "GB/s" >/dev/null
done
python - <<PY
import csv, glob, collections, os
N=$N
base=None
for dbg in (0,1,2,4,8,16,32):
    m={}
    for f in glob.glob('gpurun_out/tsec/${CORPUS:-synth}_${EXT:-1}_%d/**/*counter_collection.csv'%dbg, recursive=True):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'tamp_compress' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
        m.update({k:sum(v)/len(v) for k,v in acc.items()})
    v=m['SQ_INSTS_VALU']/N; s=m['SQ_INSTS_SALU']/N; l=m['SQ_INSTS_LDS']/N
    if dbg==0: base=(v,s,l); print('total   VALU %.0f SALU %.0f LDS %.0f per stream'%base)
    else: print('section bit %2d: VALU %+.0f SALU %+.0f LDS %+.0f'%(dbg, v-base[0], s-base[1], l-base[2]))
PY
