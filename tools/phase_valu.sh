#!/bin/bash
# Per-phase VALU / SALU wave-instructions per stream of the compress kernel, from the -DTAMP_PROF build
# (make -C tamp_amd/csrc prof): a section that may run twice without changing the result is repeated per TAMP_AMD_DBG bit
# and the difference of two `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU` runs is that section's count over all epochs of
# all streams; the truncating bits (8 / 16 / 32: return behind load / index / match + jump tables of the FIRST epoch)
# give the prologue and the first epoch's query set-up.  Writes gpurun_out/phase_valu/phase_valu.csv
#   usage (GPU box, repo root): [WLS="synth_text corpus:prose corpus:python"] [N=8192] bash tools/phase_valu.sh
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/phase_valu; rm -rf $OUT; mkdir -p $OUT
N=${N:-8192}
for WL in ${WLS:-synth_text corpus:prose corpus:python}; do
  T=${WL#corpus:}
  #      base load   index  loop wrap nodeep jump   emit   t8 t16 t32 t32+loop t32+wrap t32+jump
  DS="0 65536 131072 256 512 32768 262144 524288"
  [ "$T" = synth_text ] && DS="$DS 8 16 32 288 544 262176"
  for D in $DS; do
    WL=$WL TAMP_AMD_DBG=$D rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT -o ${T}_d$D -- timeout 200 python tools/prof_phases.py $N > $OUT/${T}_d$D.log 2>&1
  done
done
python - <<PY
import csv, collections, glob, os
N = $N
rows = []
for wl in sorted({os.path.basename(f).split('_d')[0] for f in glob.glob('$OUT/*_counter_collection.csv')}):
    val = {}
    for f in glob.glob('$OUT/%s_d*_counter_collection.csv' % wl):
        d = int(os.path.basename(f).split('_d')[1].split('_')[0])
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'tamp_compress' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        # prof_phases.py launches extended twice, then v1 twice
        for fmt, sl in (('extended', slice(0, 2)), ('v1', slice(2, 4))):
            for c in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU'):
                v = acc[c][sl]
                if v: val[(fmt, c, d)] = sum(v) / len(v) / N
    for fmt in ('extended', 'v1'):
        for c in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU'):
            g = lambda d: val.get((fmt, c, d))
            if g(0) is None: continue
            base = g(0)
            ph = collections.OrderedDict()
            ph['total'] = base
            ph['prologue + first load (up to the first barrier of the first epoch)'] = g(8)
            ph['load, all epochs'] = g(65536) - base
            ph['index (count, scan, tile scatter, query sort), all epochs'] = g(131072) - base
            ph['bucket loop, all epochs'] = g(256) - base
            ph['  of which 16-byte compares'] = base - g(32768)
            ph['wrap-zone resolution, all epochs'] = g(512) - base
            ph['jump tables (pointer doubling), all epochs'] = g(262144) - base
            ph['emit (token bits, prefix sum, scatter, HBM store), all epochs'] = g(524288) - base
            if g(8) is None:
                ph.pop('prologue + first load (up to the first barrier of the first epoch)')
                known = sum(v for k, v in ph.items() if 'all epochs' in k and not k.startswith('  '))
                ph['rest: prologue, query set-up / second pass / settled tokens of all epochs, walk, re-base'] = base - known
                for k, v in ph.items():
                    rows.append((wl, fmt, c.replace('SQ_INSTS_', ''), k, round(v)))
                continue
            first_match = g(32) - g(16)
            first_setup = first_match - (g(288) - g(32)) - (g(544) - g(32)) - (g(262176) - g(32))
            ph['first epoch only: index'] = g(16) - g(8)
            ph['first epoch only: match phase incl. jump tables'] = first_match
            ph['first epoch only: query set-up, second pass, settled tokens, epilogue (match minus loop, wrap, jump)'] = first_setup
            known = sum(ph[k] for k in ('load, all epochs', 'index (count, scan, tile scatter, query sort), all epochs', 'bucket loop, all epochs',
                                        'wrap-zone resolution, all epochs', 'jump tables (pointer doubling), all epochs',
                                        'emit (token bits, prefix sum, scatter, HBM store), all epochs'))
            ph['rest: prologue, query set-up / second pass / settled tokens of all epochs, walk, re-base'] = base - known
            for k, v in ph.items():
                rows.append((wl, fmt, c.replace('SQ_INSTS_', ''), k, round(v)))
with open('$OUT/phase_valu.csv', 'w', newline='') as fh:
    w = csv.writer(fh); w.writerow(['workload', 'format', 'counter', 'phase', 'wave_instructions_per_4KiB_stream'])
    w.writerows(rows)
for r in rows:
    if r[2] == 'VALU': print(*r)
PY
grep -h "epochs/stream\|kernel_ms" $OUT/*_d0.log
