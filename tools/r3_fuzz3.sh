# third session of the round: the final build (persistent grid, run-interior positions matched on arrival, wavefront priorities)
mkdir -p gpurun_out/r3fuzz3
FUZZ_SEED=41 timeout 300 python tools/fuzz_gpu.py 240 > gpurun_out/r3fuzz3/gpu.log 2>&1; echo rc=$? >> gpurun_out/r3fuzz3/gpu.log
FUZZ_SEED=42 timeout 200 python tools/fuzz_pieces_gpu.py 150 > gpurun_out/r3fuzz3/pieces.log 2>&1; echo rc=$? >> gpurun_out/r3fuzz3/pieces.log
FUZZ_SEED=43 timeout 150 python tools/fuzz_stream_gpu.py 90 > gpurun_out/r3fuzz3/stream.log 2>&1; echo rc=$? >> gpurun_out/r3fuzz3/stream.log
FUZZ_SEED=44 timeout 150 python tools/fuzz_encoder_resume_gpu.py 90 > gpurun_out/r3fuzz3/encres.log 2>&1; echo rc=$? >> gpurun_out/r3fuzz3/encres.log
FUZZ_SEED=45 TAMP_AMD_STATIC_GRID=1 timeout 150 python tools/fuzz_gpu.py 90 > gpurun_out/r3fuzz3/gpu_static.log 2>&1; echo rc=$? >> gpurun_out/r3fuzz3/gpu_static.log
