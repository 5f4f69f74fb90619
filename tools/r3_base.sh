set -x
mkdir -p gpurun_out/r3base
python -m pytest tests -m gpu -x -q > gpurun_out/r3base/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3base/pytest.log
python bench.py > gpurun_out/r3base/bench.json 2> gpurun_out/r3base/bench.err
python tools/realtext.py > gpurun_out/r3base/realtext.log 2>&1
for w in corpus:prose corpus:python synth_text; do echo "== $w" ; WL=$w python tools/prof_phases.py 8192; done > gpurun_out/r3base/phases.log 2>&1
