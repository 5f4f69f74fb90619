mkdir -p gpurun_out/r3l
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3l/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r3l/pytest.log
timeout 300 python tools/fuzz_gpu.py 200 > gpurun_out/r3l/fuzz.log 2>&1; echo rc=$? >> gpurun_out/r3l/fuzz.log
timeout 300 python tools/config4.py > gpurun_out/r3l/config4_full.log 2>&1
timeout 600 python bench.py > gpurun_out/r3l/bench.json 2> gpurun_out/r3l/bench.err
