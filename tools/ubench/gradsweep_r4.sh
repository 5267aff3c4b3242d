mkdir -p gpurun_out/r4u
for cfg in "1 4096" "0 4096" "1 2048" "0 2048" "1 1024"; do set -- $cfg; TOPO_AMD_GRAD_TAPER=$1 TOPO_AMD_GRAD_CHUNK_ROWS=$2 SHARD_EFF_REPS=8 timeout 400 python tools/shard_efficiency.py gradient_sigma3.25 gradient_sigma30.25 > gpurun_out/r4u/eff_t$1_c$2.json 2>/dev/null; done
timeout 600 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_halo_loopback.py -q -x -k "gradient or gate" > gpurun_out/r4u/tests.txt 2>&1
