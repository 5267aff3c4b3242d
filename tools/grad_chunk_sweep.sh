#!/bin/bash
# Gradient at 16384^2 (BASELINE config 3's own size) and 32768^2 over the chunk layout of the smooth || epilogue pipeline
# (TOPO_AMD_GRAD_CHUNKS = most chunks, TOPO_AMD_GRAD_CHUNK_ROWS = fewest rows per chunk; one process per setting: the
# settings are read once).  usage (GPU box, repo root): tools/grad_chunk_sweep.sh > gpurun_out/grad_chunk_sweep.txt
for n in 16384 32768; do
  for rows in 1024 2048 4096; do
    for nch in 4 6 8 12 16; do
      echo -n "n=$n chunk_rows=$rows max_chunks=$nch  "
      N=$n TOPO_AMD_GRAD_CHUNK_ROWS=$rows TOPO_AMD_GRAD_CHUNKS=$nch python3 tools/grad_time.py 3.25 30.25 2>&1 | tail -1
    done
  done
done
