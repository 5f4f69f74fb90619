# final validation + evidence of a round: GPU test tier, smoke(), then the rocprofv3 passes (tools/pmc_run.sh)
mkdir -p gpurun_out/r3final
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3final/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r3final/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3final/smoke.log 2>&1; echo "rc=$?" >> gpurun_out/r3final/smoke.log
TAG=${TAG:-r3c} bash tools/pmc_run.sh > gpurun_out/pmc_${TAG:-r3c}.log 2>&1
