#!/bin/bash
# rocprofv3 evidence for the compress kernel.  Separate passes: kernel-trace/stats alone; each --pmc group alone
# (FETCH_SIZE and WRITE_SIZE do not fit into one pass).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_r1j; rm -rf $OUT; mkdir -p $OUT
python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
CMD="python bench.py --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- $CMD > $OUT/stats.log 2>&1
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o write -- $CMD > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT -o sq -- $CMD > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $OUT -o sq2 -- $CMD > $OUT/sq2.log 2>&1
ls $OUT
cat $OUT/bench.json
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc_r1j/*counter_collection.csv')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'tamp_compress' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(f.split('/')[-1], {k: sum(v)/len(v) for k, v in acc.items()}, 'launches', {k: len(v) for k,v in acc.items()})
for r in csv.DictReader(open('gpurun_out/pmc_r1j/stats_kernel_stats.csv')):
    if 'tamp' in r['Name']: print(r)
PY
