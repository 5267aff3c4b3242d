#!/bin/bash
# Round-end evidence in one GPU session: PMC passes for the TPI / STD / gradient / Sx kernels, the traffic file
# bench.py reads (tied to the kernel sources it was taken on), the bench line and its kernel trace.
# usage (on the GPU box, from the repo root): tools/final_profiles.sh <git head>
R=${GRAFT_REPO_ROOT:-/root/repo}
HEAD=${1:-unknown}
F=$R/gpurun_out/final
mkdir -p $F
cd $R
PMC_SCRIPT=tools/tpi_trace.py tools/pmc_passes.sh final/pmc_tpi67 32768 67 > /dev/null 2>&1
python3 tools/traffic_from_pmc.py gpurun_out/final/pmc_tpi67 tpi_march_kernel profiles/r02_tpi67_traffic.json $HEAD > $F/traffic_tpi67.log 2>&1
cp profiles/r02_tpi67_traffic.json $F/
cp gpurun_out/final/pmc_tpi67/summary.txt $F/r02_tpi67_pmc_summary.txt
PMC_SCRIPT=tools/std_trace.py tools/pmc_passes.sh final/pmc_std67 32768 67 > /dev/null 2>&1
python3 tools/traffic_from_pmc.py gpurun_out/final/pmc_std67 std_ring_kernel $F/r02_std67_traffic.json $HEAD > $F/traffic_std67.log 2>&1
cp gpurun_out/final/pmc_std67/summary.txt $F/r02_std67_pmc_summary.txt
PMC_SCRIPT=tools/grad_trace.py tools/pmc_passes.sh final/pmc_grad30 32768 30.25 > /dev/null 2>&1
cp gpurun_out/final/pmc_grad30/summary.txt $F/r02_grad30_pmc_summary.txt
PMC_SCRIPT=tools/sx_trace.py tools/pmc_passes.sh final/pmc_sx 32768 500,2000 > /dev/null 2>&1
cp gpurun_out/final/pmc_sx/summary.txt $F/r02_sx_pmc_summary.txt
python3 bench.py > $F/r02_bench.json 2> $F/r02_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $F/trace -o bench -- python3 $R/bench.py > $F/r02_bench_under_rocprof.json 2> $F/trace.err
cp $F/trace/bench_kernel_stats.csv $F/r02_bench_kernel_stats.csv 2>/dev/null
rm -rf $R/gpurun_out/final/pmc_*/pass*/  # the raw counter files are large
ls -la $F
tail -c 300 $F/r02_bench.json
