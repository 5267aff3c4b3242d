python tools/gauss_tiling_crossover.py | grep -v "^TOPO" | awk '{print $1,$2,$3,$4,$5,$6,$7}' > gpurun_out/gt_a.txt
TOPO_AMD_GAUSS_FUSED_MAX_RADIUS=100 python tools/gauss_tiling_crossover.py | grep -v "^TOPO" | awk '{print $6,$7}' > gpurun_out/gt_b.txt
TOPO_AMD_GAUSS_FUSED_MAX_RADIUS=100 TOPO_AMD_GAUSS_FUSED_WIDE_MIN_RADIUS=24 python tools/gauss_tiling_crossover.py | grep -v "^TOPO" | awk '{print $6,$7}' > gpurun_out/gt_c.txt
echo "default(fused<=28, narrow) | fused narrow everywhere | fused wide from 24"
paste gpurun_out/gt_a.txt gpurun_out/gt_b.txt gpurun_out/gt_c.txt
