cd $GRAFT_REPO_ROOT
bash tools/ab_libs.sh base cur al32 al64 al128 2>&1 | grep -v python > gpurun_out/e9_ab.log; cat gpurun_out/e9_ab.log
