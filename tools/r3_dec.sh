mkdir -p gpurun_out/r3dec
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q -k "decod or decompress or split or round_trip or config" > gpurun_out/r3dec/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r3dec/pytest.log
timeout 300 python tools/dec_bench.py > gpurun_out/r3dec/dec_bench.log 2>&1
timeout 300 python tools/config4.py > gpurun_out/r3dec/config4.log 2>&1
timeout 300 python tools/fuzz_gpu.py 120 > gpurun_out/r3dec/fuzz.log 2>&1
