for k in 32 48 64 80; do echo "TOPO_AMD_SX_GROUP_LDS_KIB=$k"; TOPO_AMD_SX_GROUP_LDS_KIB=$k python tools/sx_multi_time.py; done > gpurun_out/sx_multi_lds.txt 2>&1
