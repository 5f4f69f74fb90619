for spw in 16 32 64; do echo "== SPW=$spw"; TAMP_AMD_SPLIT_SPW=$spw python tools/dec_traffic.py 2>&1 | grep decode; done
for sl in 14 15 16 17; do echo "== SLICE_LOG2=$sl"; TAMP_AMD_SPLIT_SLICE_LOG2=$sl python tools/dec_traffic.py 2>&1 | grep decode; done
