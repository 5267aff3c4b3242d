python -m pytest tests -m gpu -q -x -k "sx" 2>&1 | grep -E "passed|failed|Error|assert" | head
python tools/sx_time.py 2>&1 | tail -12
python tools/fuzz_sx_multi.py 40 2>&1 | tail -1
