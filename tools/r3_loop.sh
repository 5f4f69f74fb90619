mkdir -p gpurun_out/r3loop
for c in prose python; do echo "== $c"; WL=corpus:$c python tools/prof_phases.py 65536 2>&1 | grep -v amdgpu | head -6; done > gpurun_out/r3loop/phases_rt.log 2>&1
python tools/ab_time.py > gpurun_out/r3loop/ab_time.log 2>&1
TAMP_AMD_LIB=$PWD/tamp_amd/libtamp_amd_head.so python tools/ab_time.py > gpurun_out/r3loop/ab_time_head.log 2>&1
timeout 300 python tools/realtext.py > gpurun_out/r3loop/realtext.log 2>&1
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r3loop/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r3loop/pytest.log
timeout 400 python tools/fuzz_gpu.py 240 > gpurun_out/r3loop/fuzz.log 2>&1
timeout 300 python tools/fuzz_pieces_gpu.py 100 > gpurun_out/r3loop/fuzz_pieces.log 2>&1
