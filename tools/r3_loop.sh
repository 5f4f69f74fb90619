mkdir -p gpurun_out/r3loop
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r3loop/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r3loop/pytest.log
TAMP_AMD_LIB=$PWD/tamp_amd/libtamp_amd_defer0.so timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q > gpurun_out/r3loop/pytest_defer0.log 2>&1; echo "rc=$?" >> gpurun_out/r3loop/pytest_defer0.log
timeout 300 python tools/fuzz_pieces_gpu.py 100 > gpurun_out/r3loop/fuzz_pieces.log 2>&1
timeout 300 python tools/fuzz_encoder_resume_gpu.py 60 > gpurun_out/r3loop/fuzz_enc.log 2>&1
timeout 300 python tools/fuzz_stream_gpu.py 60 > gpurun_out/r3loop/fuzz_stream.log 2>&1
timeout 600 python bench.py > gpurun_out/r3loop/bench.json 2> gpurun_out/r3loop/bench.err
