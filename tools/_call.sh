cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c6
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -x -q 2>&1 | tail -3
timeout 900 bash tools/ab_libs.sh st cur > gpurun_out/c6/ab.log 2>&1; cat gpurun_out/c6/ab.log
