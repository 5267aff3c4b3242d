#!/bin/bash
# The randomised sweeps on the round's final kernels, one GPU session (~25 min): every line ends in "<n> cases, <k> failures".
R=${GRAFT_REPO_ROOT:-/root/repo}
F=$R/gpurun_out/fuzz
mkdir -p $F
cd $R
python3 tools/fuzz_parity.py 500 101 > $F/parity_default.txt 2>&1
python3 tools/fuzz_parity.py 240 103 big > $F/parity_big.txt 2>&1
python3 tools/fuzz_gradient_sx.py 400 104 > $F/gradient_sx.txt 2>&1
TOPO_AMD_GRAD_CHUNK_MIN_ROWS=64 python3 tools/fuzz_gradient_sx.py 200 105 > $F/gradient_sx_chunked.txt 2>&1
python3 tools/fuzz_sx_multi.py 200 106 > $F/sx_multi.txt 2>&1
for f in $F/parity_default.txt $F/parity_big.txt $F/gradient_sx.txt $F/gradient_sx_chunked.txt $F/sx_multi.txt; do echo "== $(basename $f)"; tail -3 $f | cut -c1-300; done
