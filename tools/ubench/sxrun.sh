for sv in 10 3; do TOPO_AMD_SX_DIAG_MIN_SAVING=$sv TOPO_AMD_SX_DIAG_MIN_COST=0 python tools/sx_time.py 2>&1 | grep -E "azimuth +(45|225)" | sed "s/^/saving>=$sv cost>=0: /"; done
