# usage: bash tools/ab_libs.sh name1 name2 ...   (libraries tamp_amd/libtamp_amd_<name>.so; "cur" = the built one), two rounds interleaved
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = cur ]; then python tools/ab_time.py; else TAMP_AMD_LIB=$PWD/tamp_amd/libtamp_amd_$v.so python tools/ab_time.py; fi
done; done 2>&1 | grep -v amdgpu
