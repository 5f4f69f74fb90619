mkdir -p gpurun_out/r3b
timeout 300 python tools/tile_check.py > gpurun_out/r3b/check.log 2>&1
timeout 300 python tools/realtext.py > gpurun_out/r3b/realtext.log 2>&1
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r3b/bench.json 2> gpurun_out/r3b/bench.err
