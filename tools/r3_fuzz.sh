mkdir -p gpurun_out/r3fuzz
timeout 700 python tools/fuzz_pieces_gpu.py 600 > gpurun_out/r3fuzz/pieces.log 2>&1; echo rc=$? >> gpurun_out/r3fuzz/pieces.log
timeout 400 python tools/fuzz_gpu.py 300 > gpurun_out/r3fuzz/gpu.log 2>&1; echo rc=$? >> gpurun_out/r3fuzz/gpu.log
timeout 300 python tools/fuzz_stream_gpu.py 180 > gpurun_out/r3fuzz/stream.log 2>&1; echo rc=$? >> gpurun_out/r3fuzz/stream.log
timeout 300 python tools/fuzz_encoder_resume_gpu.py 180 > gpurun_out/r3fuzz/encres.log 2>&1; echo rc=$? >> gpurun_out/r3fuzz/encres.log
TAMP_AMD_ENCODER=tile timeout 400 python tools/fuzz_gpu.py 300 > gpurun_out/r3fuzz/gpu_tile.log 2>&1; echo rc=$? >> gpurun_out/r3fuzz/gpu_tile.log
