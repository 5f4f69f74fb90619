mkdir -p gpurun_out/r3final
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3final/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r3final/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3final/smoke.log 2>&1; echo "rc=$?" >> gpurun_out/r3final/smoke.log
TAG=r3b bash tools/pmc_run.sh > gpurun_out/pmc_r3b.log 2>&1
