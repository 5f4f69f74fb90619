mkdir -p gpurun_out/r3loop
bash tools/ab_libs.sh head cur defer12 defer48 > gpurun_out/r3loop/ab_prio.log 2>&1
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r3loop/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r3loop/pytest.log
timeout 300 python tools/fuzz_gpu.py 200 > gpurun_out/r3loop/fuzz.log 2>&1
timeout 300 python tools/fuzz_pieces_gpu.py 80 > gpurun_out/r3loop/fuzz_pieces.log 2>&1
timeout 300 python tools/fuzz_stream_gpu.py 50 > gpurun_out/r3loop/fuzz_stream.log 2>&1
timeout 300 python tools/fuzz_encoder_resume_gpu.py 50 > gpurun_out/r3loop/fuzz_enc.log 2>&1
timeout 600 python bench.py > gpurun_out/r3loop/bench.json 2> gpurun_out/r3loop/bench.err
timeout 300 python tools/config5.py > gpurun_out/r3loop/config5.log 2>&1
