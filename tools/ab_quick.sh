# usage: bash tools/ab_quick.sh name1 name2 ...   -- same-box A/B of tuning builds (tamp_amd/libtamp_amd_<name>.so, "cur" = the product
# build): a compress-parity subset of the GPU tier on every variant (PARITY=full: every compress test incl. block mode and lazy
# matching), then the kernel times of tools/ab_time.py, two rounds interleaved.
for v in "$@"; do
  [ "$v" = cur ] && continue
  echo "== parity $v"
  if [ "${PARITY:-}" = full ]; then
    TAMP_AMD_LIB=$PWD/tamp_amd/libtamp_amd_$v.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_round5.py tests/test_gpu_round6.py -m gpu -x -q 2>&1 | tail -2
  else
    TAMP_AMD_LIB=$PWD/tamp_amd/libtamp_amd_$v.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q -k "known_answer or generated_reference or differential_vs_oracle or run_aware or long_runs or crowded or real_text or fuzz" 2>&1 | tail -2
  fi
done
bash tools/ab_libs.sh "$@"
