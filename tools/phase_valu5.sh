#!/bin/bash
# Round 5: per-phase VALU / SALU wave-instructions per 4 KiB stream of the compress kernel, from the -DTAMP_PROF build
# (make -C tamp_amd/csrc prof).  Every row is the difference of two `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU` runs in which
# ONE section ran twice WITHOUT changing what the kernel computes (TAMP_AMD_DBG bit), so no row can come out negative:
#   0x10000 load   0x20000 index   0x100 bucket loop   0x1000000 its 16-byte compares (re-executed, same result)   0x200 wrap zone
#   0x2000000 second pass (run list)   0x4000000 settled tokens (a dry run in front)   0x40000 jump tables   0x80000 emit
#   0x8000000 the walk's main loop (twice from the same state; waves 1-3 serve both runs' searches)
#   0x18000000 ... with the second run's token listing skipped (difference to 0x8000000 = the listing)
#   8 = return behind the first epoch's load (prologue + first load)
# Writes gpurun_out/phase_valu5/phase_valu.csv.   usage (GPU box): [WLS=...] [N=8192] bash tools/phase_valu5.sh
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/phase_valu5; rm -rf $OUT; mkdir -p $OUT
N=${N:-8192}
for WL in ${WLS:-synth_text corpus:prose corpus:markup corpus:python}; do
  T=${WL#corpus:}
  for D in 0 65536 131072 256 16777216 512 33554432 67108864 262144 524288 134217728 402653184 8; do
    WL=$WL TAMP_AMD_DBG=$D timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT -o ${T}_d$D -- python tools/prof_phases.py $N > $OUT/${T}_d$D.log 2>&1 < /dev/null
  done
done
python - <<PY
import csv, collections, glob, os
N = $N
rows = []
for wl in ['synth_text', 'prose', 'markup', 'python']:
    val = {}
    for f in glob.glob('$OUT/%s_d*_counter_collection.csv' % wl):
        d = int(os.path.basename(f).split('_d')[1].split('_')[0])
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'tamp_compress' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for fmt, sl in (('extended', slice(0, 2)), ('v1', slice(2, 4))):   # prof_phases.py: extended twice, then v1 twice
            for c in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU'):
                v = acc[c][sl]
                if v: val[(fmt, c, d)] = sum(v) / len(v) / N
    epochs = {}
    try:
        import re
        txt = open('$OUT/%s_d0.log' % wl).read()
        e = re.findall(r'epochs/stream=([0-9.]+)', txt)
        epochs = {'extended': float(e[0]), 'v1': float(e[1])}
    except Exception:
        pass
    for fmt in ('extended', 'v1'):
        for c in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU'):
            g = lambda d: val.get((fmt, c, d))
            if g(0) is None: continue
            base = g(0)
            diff = lambda d: (g(d) - base) if g(d) is not None else None
            ph = collections.OrderedDict()
            ph['total'] = base
            load = diff(65536)
            ph['load, all epochs'] = load
            ph['index (count, scan, tile scatter, query sort), all epochs'] = diff(131072)
            ph['bucket loop, all epochs'] = diff(256)
            ph['  of which 16-byte compares (re-executed)'] = diff(16777216)
            ph['wrap-zone resolution, all epochs'] = diff(512)
            ph['second pass over the run list, all epochs'] = diff(33554432)
            ph['settled tokens (dry run), all epochs'] = diff(67108864)
            ph['jump tables (pointer doubling), all epochs'] = diff(262144)
            walk = diff(134217728)
            ph['walk: main loop incl. slow steps, searches served by all four wavefronts'] = walk
            if walk is not None and g(402653184) is not None:
                ph['  of which token listing'] = g(134217728) - g(402653184)
            ph['emit (token bits, prefix sum, scatter, HBM store), all epochs'] = diff(524288)
            if g(8) is not None and load is not None and epochs.get(fmt):
                ph['prologue (window, bit buffer, tables; per stream)'] = g(8) - load / epochs[fmt]
            known = sum(v for k, v in ph.items() if k != 'total' and not k.startswith('  ') and v is not None)
            ph['rest: query set-up and result stores of the match phase, on-demand matches, re-base, epoch bookkeeping'] = base - known
            for k, v in ph.items():
                if v is not None: rows.append((wl, fmt, c.replace('SQ_INSTS_', ''), k, round(v), ('%.1f%%' % (100.0 * v / base))))
with open('$OUT/phase_valu.csv', 'w', newline='') as fh:
    w = csv.writer(fh); w.writerow(['workload', 'format', 'counter', 'phase', 'wave_instructions_per_4KiB_stream', 'share_of_total'])
    w.writerows(rows)
for r in rows:
    if r[2] == 'VALU' and r[1] == 'extended': print(*r)
PY
grep -h "epochs/stream\|kernel_ms" $OUT/*_d0.log | head -16
