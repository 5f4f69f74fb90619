mkdir -p gpurun_out/r3h
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3h/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r3h/pytest.log
timeout 600 python bench.py > gpurun_out/r3h/bench.json 2> gpurun_out/r3h/bench.err
timeout 300 python tools/config5.py > gpurun_out/r3h/config5.log 2>&1
timeout 300 python tools/short_msgs.py > gpurun_out/r3h/short.log 2>&1
