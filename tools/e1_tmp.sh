cd $GRAFT_REPO_ROOT
bash tools/ab_libs.sh al0 cur al128 al512 2>&1 | grep -v "v1" > gpurun_out/e13_ab.log; cat gpurun_out/e13_ab.log
