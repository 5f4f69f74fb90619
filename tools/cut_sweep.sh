for rep in 1 2; do for v in 6 2 3 4; do echo "cut_run $v"; TAMP_AMD_CUT_RUN=$v python tools/ab_time.py 2>&1 | grep -E "prose|python" | sed 's/libtamp_amd.so//'; done; done
bash tools/ab_libs.sh cur lr5 lr6 lr12
