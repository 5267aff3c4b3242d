python -m pytest tests -m gpu -q -x -k "sx" 2>&1 | grep -E "passed|failed|Error|assert" | head
for f in 1 0; do TOPO_AMD_SX_PAIRS=$f python tools/sx_time.py 2>&1 | tail -12 | sed "s/^/pairs=$f: /"; done
python tools/fuzz_sx_multi.py 60 2>&1 | tail -1
python tools/fuzz_gradient_sx.py 90 17 2>&1 | tail -1
