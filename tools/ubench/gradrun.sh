for cfg in "16 2048" "8 4096" "12 2560"; do set -- $cfg; TOPO_AMD_GRAD_CHUNKS=$1 TOPO_AMD_GRAD_CHUNK_ROWS=$2 python tools/grad_time.py 3.25 30.25 | sed "s/^/chunks $1 rows>=$2: /"; done
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "nan" 2>&1 | tail -2
