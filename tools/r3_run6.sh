mkdir -p gpurun_out/r3i
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q > gpurun_out/r3i/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r3i/pytest.log
timeout 300 python tools/realtext.py > gpurun_out/r3i/realtext.log 2>&1
for w in corpus:prose corpus:python; do echo "== $w"; WL=$w python tools/prof_phases.py 8192; done > gpurun_out/r3i/phases.log 2>&1
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r3i/bench.json 2> gpurun_out/r3i/bench.err
