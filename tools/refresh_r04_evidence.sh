R=${GRAFT_REPO_ROOT:-/root/repo}
F=$R/gpurun_out/final4
mkdir -p $F
cd $R
PMC_SCRIPT=tools/grad_trace.py tools/pmc_passes.sh final4/pmc_grad30 32768 30.25 > /dev/null 2>&1
cp gpurun_out/final4/pmc_grad30/summary.txt $F/r04_grad30_pmc_summary.txt
TOPO_AMD_HALO_LOOPBACK=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --ny 4096 > $F/r04_bench_loopback_4096rows.json 2> $F/loopback.err
python3 bench.py > $F/r04_bench.json 2> $F/r04_bench.err
SHARD_EFF_REPS=8 timeout 600 python3 tools/shard_efficiency.py > $F/r04_shard_efficiency.json 2> $F/eff.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $F/trace -o bench -- python3 $R/bench.py --no-end-to-end > $F/r04_bench_under_rocprof.json 2> $F/trace.err
cp $F/trace/bench_kernel_stats.csv $F/r04_bench_kernel_stats.csv 2>/dev/null
rm -rf $R/gpurun_out/final4/pmc_*/pass*/ $F/trace
ls -la $F
