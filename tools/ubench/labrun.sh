cd tools/ubench
for s in 7 11 13 17; do
  for dem in 0 1; do
    TOPO_AMD_TPI_RING_MAX=17 ./tpi_lab_b$s.bin any_tpi 32768 5 $dem 2>&1 | tail -1 | sed "s/^/both size $s /"
    TOPO_AMD_TPI_RING_MAX=17 TOPO_AMD_TPI_RING_BOTH=0 ./tpi_lab_b$s.bin any_tpi 32768 5 $dem 2>&1 | tail -1 | sed "s/^/old  size $s /"
  done
done
TOPO_AMD_TPI_RING_MAX=11 ./tpi_lab_b13.bin any_tpi 32768 5 0 2>&1 | tail -1 | sed "s/^/march size 13 /"
TOPO_AMD_TPI_RING_MAX=11 ./tpi_lab_b17.bin any_tpi 32768 5 0 2>&1 | tail -1 | sed "s/^/march size 17 /"
TOPO_AMD_TPI_RING_MAX=11 ./tpi_lab_b13.bin any_tpi 32768 5 1 2>&1 | tail -1 | sed "s/^/march size 13 /"
TOPO_AMD_TPI_RING_MAX=11 ./tpi_lab_b17.bin any_tpi 32768 5 1 2>&1 | tail -1 | sed "s/^/march size 17 /"
