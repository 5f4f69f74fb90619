mkdir -p gpurun_out/r3loop
bash tools/ab_libs.sh prev cur > gpurun_out/r3loop/ab_filter.log 2>&1
for c in prose python; do echo "== $c"; WL=corpus:$c python tools/prof_phases.py 65536 2>&1 | grep -v amdgpu | head -6; done > gpurun_out/r3loop/phases_rt.log 2>&1
timeout 300 python tools/realtext.py > gpurun_out/r3loop/realtext.log 2>&1
timeout 300 python tools/fuzz_gpu.py 150 > gpurun_out/r3loop/fuzz.log 2>&1
timeout 300 python tools/fuzz_pieces_gpu.py 60 > gpurun_out/r3loop/fuzz_pieces.log 2>&1
