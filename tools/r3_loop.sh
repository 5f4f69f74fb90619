mkdir -p gpurun_out/r3loop
for b in 1536 1280 1024 1792 2048; do echo "blk $b: $(TAMP_AMD_BLK=$b python tools/ab_time.py 2>&1 | grep -E 'synthetic ext|prose|python' | sed 's/libtamp_amd.so//' | tr '\n' '|')"; done > gpurun_out/r3loop/blk.log 2>&1
