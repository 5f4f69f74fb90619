set -x
mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3a/pytest.log
timeout 300 python tools/realtext.py > gpurun_out/r3a/realtext.log 2>&1
TAMP_AMD_ENCODER=epoch timeout 300 python tools/realtext.py > gpurun_out/r3a/realtext_epoch.log 2>&1
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r3a/prof -- python /root/repo/tools/realtext.py > /root/repo/gpurun_out/r3a/prof.log 2>&1
