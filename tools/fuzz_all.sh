#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
F=$R/gpurun_out/fuzz
mkdir -p $F
cd $R
python3 tools/fuzz_parity.py 500 101 > $F/parity_default.txt 2>&1
TOPO_AMD_STD_RING_MIN=999 TOPO_AMD_TPI_RING_MIN=999 python3 tools/fuzz_parity.py 240 102 > $F/parity_noring.txt 2>&1
python3 tools/fuzz_parity.py 240 103 big > $F/parity_big.txt 2>&1
python3 tools/fuzz_gradient_sx.py 500 104 > $F/gradient_sx.txt 2>&1
TOPO_AMD_GRAD_CHUNK_MIN_ROWS=64 python3 tools/fuzz_gradient_sx.py 240 105 > $F/gradient_sx_chunked.txt 2>&1
python3 tools/fuzz_sx_multi.py 400 106 > $F/sx_multi.txt 2>&1
PMC_SCRIPT=tools/grad_trace.py tools/pmc_passes.sh fuzz/pmc_grad325 32768 3.25 > /dev/null 2>&1
cp gpurun_out/fuzz/pmc_grad325/summary.txt $F/r02_grad325_pmc_summary.txt
rm -rf gpurun_out/fuzz/pmc_grad325/pass*/
for f in $F/*.txt; do echo "== $f"; tail -1 $f | cut -c1-300; done
