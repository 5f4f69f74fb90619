mkdir -p gpurun_out/r3c
timeout 300 python tools/tile_check.py > gpurun_out/r3c/check.log 2>&1
timeout 300 python tools/realtext.py > gpurun_out/r3c/realtext.log 2>&1
CORPUS=synth EXT=1 bash tools/tile_sections.sh > gpurun_out/r3c/tsec_synth.log 2>&1
