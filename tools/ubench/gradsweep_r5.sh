#!/bin/bash
# gradient on 32768^2, sigma 3.25 / 30.25, for chunk counts x chunk rows x taper (does the epilogue find
# the smoothed chunk in the 256 MiB Infinity Cache when a chunk's plane is 128 MiB = 1024 rows?)
mkdir -p gpurun_out
for cfg in "8 4096 1" "16 2048 1" "32 1024 1" "32 1024 0" "64 512 1" "24 1408 1"; do
  set -- $cfg
  echo "chunks<=$1 rows>=$2 taper=$3: $(TOPO_AMD_GRAD_CHUNKS=$1 TOPO_AMD_GRAD_CHUNK_ROWS=$2 TOPO_AMD_GRAD_TAPER=$3 timeout 300 python tools/grad_time.py 3.25 30.25 2>&1 | tail -1)"
done
