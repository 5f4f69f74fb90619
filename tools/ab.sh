for rep in 1 2; do for v in $VARS; do TAMP_VAR=$v python tools/ab_bench.py 2>&1 | grep -v amdgpu.ids; done; done
