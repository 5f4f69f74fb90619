mkdir -p gpurun_out/r3loop
timeout 600 python tools/xcd_probe.py > gpurun_out/r3loop/xcd_probe.log 2>&1
timeout 600 python bench.py > gpurun_out/r3loop/bench.json 2> gpurun_out/r3loop/bench.err
timeout 300 python tools/config5.py > gpurun_out/r3loop/config5.log 2>&1
timeout 300 python tools/fuzz_gpu.py 120 > gpurun_out/r3loop/fuzz.log 2>&1
