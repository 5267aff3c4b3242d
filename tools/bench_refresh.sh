#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
F=$R/gpurun_out/final
mkdir -p $F
cd $R
python3 bench.py > $F/r03_bench.json 2> $F/r03_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf $F/trace
rocprofv3 --kernel-trace --stats --output-format csv -d $F/trace -o bench -- python3 $R/bench.py > $F/r03_bench_under_rocprof.json 2> $F/trace.err
cp $F/trace/bench_kernel_stats.csv $F/r03_bench_kernel_stats.csv
cd $R && python3 tools/parity_report.py > $F/parity_report.txt 2>&1
tail -c 200 $F/r03_bench.json
