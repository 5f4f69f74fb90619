cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/e4_bench.json 2> gpurun_out/e4_bench.err
tail -3 gpurun_out/e4_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/e4_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline'])
print(d['config'].get('real_text_MBps'))
print(d.get('also',{}).get('decompress'))
PY
