mkdir -p gpurun_out/r3j
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3j/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r3j/pytest.log
timeout 400 python tools/fuzz_gpu.py 240 > gpurun_out/r3j/fuzz.log 2>&1; echo rc=$? >> gpurun_out/r3j/fuzz.log
timeout 300 python tools/dec_bench.py > gpurun_out/r3j/dec_bench.log 2>&1
timeout 300 python tools/config4.py 262144 > gpurun_out/r3j/config4.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r3j -o dec2_fetch -- python tools/dec_traffic.py > gpurun_out/r3j/dec2_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r3j -o dec2_write -- python tools/dec_traffic.py > gpurun_out/r3j/dec2_write.log 2>&1
