for g in 4 8 16; do echo "A_GROUP $g"; SDP_COL_A_GROUP=$g timeout 100 python tools/filter_ab.py 256 2>&1 | grep "True: kernel"; done
for lw in 16 32; do echo "A_LW $lw"; SDP_COL_A_LW=$lw timeout 100 python tools/filter_ab.py 256 2>&1 | grep "True: kernel"; done
