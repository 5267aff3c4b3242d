cd tools/ubench
for s in 7 17 31 41; do
  for v in any_std any_tpi_std; do
    for dem in 0 1; do
      ./tpi_lab_b$s.bin $v 32768 5 $dem 2>&1 | tail -1 | sed "s/^/both size $s /"
      TOPO_AMD_STD_RING_BOTH=0 ./tpi_lab_b$s.bin $v 32768 5 $dem 2>&1 | tail -1 | sed "s/^/old  size $s /"
    done
  done
done
