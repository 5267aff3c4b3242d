#!/bin/bash
# Round-end evidence in one GPU session: PMC passes for the TPI / STD kernels, the vector-ALU bound, the traffic file
# bench.py reads (tied to the kernel sources it was taken on), the bench line and its kernel trace.
# usage (on the GPU box, from the repo root): tools/final_profiles.sh <git head>
R=${GRAFT_REPO_ROOT:-/root/repo}
HEAD=${1:-unknown}
F=$R/gpurun_out/final
mkdir -p $F
cd $R
PMC_SCRIPT=tools/tpi_trace.py tools/pmc_passes.sh final/pmc_tpi67 32768 67 int > /dev/null 2>&1
python3 tools/traffic_from_pmc.py gpurun_out/final/pmc_tpi67 tpi_march_kernel profiles/r06_tpi67_traffic.json $HEAD > $F/traffic_tpi67.log 2>&1
cp profiles/r06_tpi67_traffic.json $F/
cp gpurun_out/final/pmc_tpi67/summary.txt $F/r06_tpi67_pmc_summary.txt
cp gpurun_out/final/pmc_tpi67/summary.txt profiles/r06_tpi67_pmc_summary.txt
python3 tools/valu_bound.py > profiles/r06_tpi67_valu_bound.json 2> $F/valu_bound.err
cp profiles/r06_tpi67_valu_bound.json $F/
# the scaled one-chain route on fractional elevations (tpi_scaled_march_kernel<67, 60, 12, true> from the second call on)
PMC_SCRIPT=tools/tpi_trace.py tools/pmc_passes.sh final/pmc_tpi67_frac 32768 67 frac > /dev/null 2>&1
cp gpurun_out/final/pmc_tpi67_frac/summary.txt $F/r06_tpi67_fractional_pmc_summary.txt
PMC_SCRIPT=tools/std_trace.py tools/pmc_passes.sh final/pmc_std67 32768 67 > /dev/null 2>&1
# one traffic file per kernel variant (STD alone, TPI + STD): round 3's file averaged the two
python3 tools/traffic_from_pmc.py gpurun_out/final/pmc_std67 "std_ring_kernel<67, false" $F/r06_std67_traffic.json $HEAD > $F/traffic_std67.log 2>&1
python3 tools/traffic_from_pmc.py gpurun_out/final/pmc_std67 "std_ring_kernel<67, true" $F/r06_tpi_std67_traffic.json $HEAD > $F/traffic_tpi_std67.log 2>&1
cp gpurun_out/final/pmc_std67/summary.txt $F/r06_std67_pmc_summary.txt
cp gpurun_out/final/pmc_std67/summary.txt profiles/r06_std67_pmc_summary.txt
python3 tools/valu_bound.py std > profiles/r06_std67_valu_bound.json 2> $F/valu_bound_std.err
cp profiles/r06_std67_valu_bound.json $F/
# the small discs' kernel with staging waves apart from chain waves (std_ring_spec_kernel<7, .>)
PMC_SCRIPT=tools/std_trace.py tools/pmc_passes.sh final/pmc_std7 32768 7 > /dev/null 2>&1
cp gpurun_out/final/pmc_std7/summary.txt $F/r06_std7_pmc_summary.txt
cp gpurun_out/final/pmc_std7/summary.txt profiles/r06_std7_pmc_summary.txt
python3 tools/traffic_from_pmc.py gpurun_out/final/pmc_std7 "std_ring_spec_kernel<7, false" $F/r06_std7_traffic.json $HEAD > $F/traffic_std7.log 2>&1
# the gradient at both sigmas of config 3 (round 3: f16 matrix pipe, fused short filters)
PMC_SCRIPT=tools/grad_trace.py tools/pmc_passes.sh final/pmc_grad325 32768 3.25 > /dev/null 2>&1
cp gpurun_out/final/pmc_grad325/summary.txt $F/r06_grad325_pmc_summary.txt
PMC_SCRIPT=tools/grad_trace.py tools/pmc_passes.sh final/pmc_grad30 32768 30.25 > /dev/null 2>&1
cp gpurun_out/final/pmc_grad30/summary.txt $F/r06_grad30_pmc_summary.txt
# the valley index on the matrix pipe (round 6): MFMA busy cycles next to the vector ALU's and the LDS's
PMC_EXTRA="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES" PMC_SCRIPT=tools/valley_trace.py tools/pmc_passes.sh final/pmc_valley 16384 7 > /dev/null 2>&1
cp gpurun_out/final/pmc_valley/summary.txt $F/r06_valley_pmc_summary.txt
# (the Sx kernels' counters: profiles/r02_sx_pmc_summary.txt stands for the axis-aligned scans; PMC_SCRIPT=tools/sx_trace.py re-takes them)
# the sharded step with the real exchange on one GPU: one 4096-row shard of the 8-GPU split, neighbours = itself
TOPO_AMD_HALO_LOOPBACK=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --ny 4096 > $F/r06_bench_loopback_4096rows.json 2> $F/loopback.err
python3 bench.py > $F/r06_bench.json 2> $F/r06_bench.err
cd /tmp && export TMPDIR=/tmp
# (--no-end-to-end: that section launches tpi_march_kernel<67> on a 16384^2 DEM, which would enter the kernel's average)
rocprofv3 --kernel-trace --stats --output-format csv -d $F/trace -o bench -- python3 $R/bench.py --no-end-to-end > $F/r06_bench_under_rocprof.json 2> $F/trace.err
cp $F/trace/bench_kernel_stats.csv $F/r06_bench_kernel_stats.csv 2>/dev/null
rm -rf $R/gpurun_out/final/pmc_*/pass*/  # the raw counter files are large
ls -la $F
tail -c 300 $F/r06_bench.json
