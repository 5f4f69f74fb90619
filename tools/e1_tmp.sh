cd $GRAFT_REPO_ROOT
bash tools/ab_libs.sh cur wz 2>&1 | grep -v "synthetic v1" > gpurun_out/e23_ab.log; cat gpurun_out/e23_ab.log
timeout 600 python -m pytest tests -m gpu -x -q -k "real_text or corpus or markup or run_aware or differential" 2>&1 | tail -2
