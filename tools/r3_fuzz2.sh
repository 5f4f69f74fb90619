mkdir -p gpurun_out/r3fuzz2
FUZZ_SEED=31 timeout 700 python tools/fuzz_pieces_gpu.py 600 > gpurun_out/r3fuzz2/pieces.log 2>&1; echo rc=$? >> gpurun_out/r3fuzz2/pieces.log
FUZZ_SEED=32 timeout 700 python tools/fuzz_gpu.py 600 > gpurun_out/r3fuzz2/gpu.log 2>&1; echo rc=$? >> gpurun_out/r3fuzz2/gpu.log
FUZZ_SEED=33 timeout 400 python tools/fuzz_resume_gpu.py 300 > gpurun_out/r3fuzz2/decres.log 2>&1; echo rc=$? >> gpurun_out/r3fuzz2/decres.log
FUZZ_SEED=34 timeout 400 python tools/fuzz_stream_gpu.py 300 > gpurun_out/r3fuzz2/stream.log 2>&1; echo rc=$? >> gpurun_out/r3fuzz2/stream.log
