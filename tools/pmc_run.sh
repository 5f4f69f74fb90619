#!/bin/bash
# rocprofv3 evidence (rounds 2-3).  Separate passes: kernel-trace/stats alone; each --pmc group alone (FETCH_SIZE and WRITE_SIZE
# do not fit into one pass).  Summaries land in gpurun_out/pmc_$TAG; copy what is to be tracked into profiles/.
#   usage (GPU box, repo root): TAG=r2a bash tools/pmc_run.sh
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
TAG=${TAG:-r6}
OUT=gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
# ---- compress kernel (configs[1], the bench command) ----
CMD="python bench.py --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- $CMD > $OUT/stats.log 2>&1
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o write -- $CMD > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT -o sq -- $CMD > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $OUT -o sq2 -- $CMD > $OUT/sq2.log 2>&1
# ---- decoders: BASELINE configs[3] at full size (1,048,576 streams, windows 2^8..2^12) ----
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o dec4_stats -- python tools/config4.py > $OUT/dec4_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o dec4_fetch -- python tools/config4.py > $OUT/dec4_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o dec4_write -- python tools/config4.py > $OUT/dec4_write.log 2>&1
# ---- split decoder on the bench batch: instruction counters per kernel ----
N=65536 bash tools/dec_split_pmc.sh > $OUT/dec_split_pmc.log 2>&1
cp gpurun_out/sp/p_counter_collection.csv $OUT/dec_split_sq_counter_collection.csv 2>/dev/null
cp gpurun_out/sp/s_kernel_stats.csv $OUT/dec_split_kernel_stats.csv 2>/dev/null
# ---- BASELINE configs[4] share and 256-byte messages without padding ----
python tools/config5.py > $OUT/config5.log 2>&1
python tools/short_msgs.py 1048576 > $OUT/short_msgs.log 2>&1
# ---- decoders on the bench batch (65,536 x 4 KiB, w=10), all four ----
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o dec2_stats -- python tools/dec_bench.py > $OUT/dec2_stats.log 2>&1
# ---- real text, both formats ----
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o realtext_stats -- python tools/realtext.py > $OUT/realtext_stats.log 2>&1
# ---- real text (frozen corpora, extended format) and configs[4] messages: instruction counters (round 3) ----
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
for c in prose markup python; do
  CORPUS=$c EXT=1 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT -o realtext_${c}_sq -- python tools/one_corpus.py 32768 > $OUT/realtext_${c}_sq.log 2>&1
done
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT -o c5_sq -- python tools/config5.py > $OUT/c5_sq.log 2>&1
# ---- decode of the bench batch (65,536 x 4 KiB, split decoder): HBM traffic ----
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o dec2_fetch -- python tools/dec_traffic.py > $OUT/dec2_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o dec2_write -- python tools/dec_traffic.py > $OUT/dec2_write.log 2>&1
# ---- ONE long stream decoded by the whole device (round 6): v1 and the extended format ----
EXT=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o longdec_v1 -- python tools/long_dec_prof.py > $OUT/longdec_v1.log 2>&1
EXT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o longdec_ext -- python tools/long_dec_prof.py > $OUT/longdec_ext.log 2>&1
ls $OUT
cat $OUT/bench.json
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob('$OUT/*counter_collection.csv')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'tamp_' in r['Kernel_Name'] and 'header_scan' not in r['Kernel_Name']:
            acc[(r['Kernel_Name'][:60], r['Counter_Name'])].append(float(r['Counter_Value']))
    print(f.split('/')[-1], {k: (round(sum(v)/len(v)), len(v)) for k, v in acc.items()})
for f in sorted(glob.glob('$OUT/*_kernel_stats.csv')):
    for r in csv.DictReader(open(f)):
        if 'tamp' in r['Name']: print(f.split('/')[-1], r['Name'][:70], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
grep -h "GB/s\|per stream" $OUT/longdec_v1.log $OUT/longdec_ext.log $OUT/dec4_stats.log $OUT/dec2_stats.log $OUT/realtext_stats.log $OUT/config5.log $OUT/short_msgs.log $OUT/dec_split_pmc.log
