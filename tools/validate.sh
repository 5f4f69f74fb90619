#!/bin/bash
# Validation and evidence of a round on one MI355X box (repo root; results under gpurun_out/).
#   bash tools/validate.sh fuzz  [SEED=50] [SCALE=1]   randomised differential runs against the reference C / oracle:
#                                                      batch kernels, pieces, streaming scripts, both resume kernels, static grid, block mode
#   TAG=r4 bash tools/validate.sh final                GPU test tier, smoke(), then the rocprofv3 passes of tools/pmc_run.sh
# (one maintained script instead of the per-round one-offs of rounds 1-3)
cd "${GRAFT_REPO_ROOT:?}"
MODE=${1:-fuzz}
if [ "$MODE" = fuzz ]; then
  S=${SEED:-50}; K=${SCALE:-1}; OUT=gpurun_out/fuzz_$S; mkdir -p $OUT
  run() { name=$1; secs=$2; shift 2; FUZZ_SEED=$S timeout $((secs + 90)) "$@" $secs > $OUT/$name.log 2>&1; echo rc=$? >> $OUT/$name.log; S=$((S + 1)); }
  run gpu $((240 * K)) python tools/fuzz_gpu.py
  run pieces $((150 * K)) python tools/fuzz_pieces_gpu.py
  run stream $((90 * K)) python tools/fuzz_stream_gpu.py
  run encres $((90 * K)) python tools/fuzz_encoder_resume_gpu.py
  run decres $((90 * K)) python tools/fuzz_resume_gpu.py
  TAMP_AMD_STATIC_GRID=1 run gpu_static $((60 * K)) python tools/fuzz_gpu.py
  run block $((90 * K)) python tools/fuzz_block_gpu.py
  run longdec $((90 * K)) python tools/fuzz_long_decode_gpu.py
  for f in $OUT/*.log; do echo "[$(basename $f .log)] $(grep -h 'fuzz\|rc=' $f | tail -2 | tr '\n' ' ')"; done | tee $OUT/summary.txt
else
  TAG=${TAG:-r4}; OUT=gpurun_out/final_$TAG; mkdir -p $OUT
  timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
  python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "rc=$?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
  TAG=$TAG bash tools/pmc_run.sh > gpurun_out/pmc_$TAG.log 2>&1; tail -40 gpurun_out/pmc_$TAG.log
fi
