# usage: bash tools/ab_scaling.sh name1 name2 ...  -- tools/ab_time.py and the strong-scaling prediction for tuning builds
for v in "$@"; do
  export TAMP_AMD_LIB=$PWD/tamp_amd/libtamp_amd_$v.so
  python tools/ab_time.py 2>&1 | grep -v amdgpu
  python tools/strong_scaling.py 5 2>&1 | grep -E "predicted|t_ms_all|\"t_ms\"" | tr -d '\n'; echo " [$v]"
done
