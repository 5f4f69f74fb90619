cd $GRAFT_REPO_ROOT
bash tools/ab_libs.sh base cur > gpurun_out/e11_ab.log 2>&1; cat gpurun_out/e11_ab.log
