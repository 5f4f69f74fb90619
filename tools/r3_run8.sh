mkdir -p gpurun_out/r3k
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q -k "decod or decompress or split or round_trip or config" > gpurun_out/r3k/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r3k/pytest.log
timeout 300 python tools/dec_bench.py > gpurun_out/r3k/dec_bench.log 2>&1
timeout 300 python tools/config4.py 262144 > gpurun_out/r3k/config4.log 2>&1
