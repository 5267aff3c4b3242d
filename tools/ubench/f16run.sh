for e in 1024 0 512 2048; do TOPO_AMD_GRAD_EDGE_CHUNK_ROWS=$e python tools/grad_time.py 3.25 30.25 | sed "s/^/edge chunk rows $e: /"; done
python tools/grad_chunk_check.py 2>&1 | tail -2
python -m pytest tests/test_gpu_blocks.py -m gpu -q -x -k "gradient or config3 or config5" 2>&1 | grep -E "passed|failed|Error|assert" | head -3
