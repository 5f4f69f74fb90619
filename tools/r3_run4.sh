mkdir -p gpurun_out/r3d
timeout 1500 python -m pytest tests/test_gpu_round3.py -m gpu -x -q --durations=10 > gpurun_out/r3d/pytest3.log 2>&1; echo "rc=$?" >> gpurun_out/r3d/pytest3.log
timeout 600 python bench.py > gpurun_out/r3d/bench.json 2> gpurun_out/r3d/bench.err
