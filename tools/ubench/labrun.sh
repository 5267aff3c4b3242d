cd tools/ubench
for b in a0 a1; do for v in any_std any_tpi_std; do ./tpi_lab_$b.bin $v 32768 5 2>&1 | tail -1 | sed "s/^/$b /"; done; done
for b in s31a0 s31a1 s7a0 s7a1; do for v in any_tpi any_std any_tpi_std; do ./tpi_lab_$b.bin $v 32768 5 2>&1 | tail -1 | sed "s/^/$b /"; done; done
