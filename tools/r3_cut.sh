for c in 2 3 4 6 10; do echo "== TAMP_AMD_CUT_RUN=$c"; TAMP_AMD_CUT_RUN=$c python tools/realtext.py 2>&1 | grep "GB/s" | grep " ext"; done
