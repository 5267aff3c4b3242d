#!/bin/bash
# First contact with a multi-GPU node: the N > 1 branch of the ghost-row exchange has only ever run
# in loop-back on one GPU.  One command; prints, for N = 2, 4, 8 ranks, the bench line (value, ms per step, per-descriptor
# shard table), the blocks that gave up at the ghost-row gate, and how many channels RCCL really used for the neighbour
# sends (the reserve-one-CU-per-XCD design assumes NCCL_MAX_NCHANNELS=8 is honoured on the xGMI P2P path).
#
#   tools/first_contact.sh [steps=20]
#
# Order: careful gate first (the default: correct whatever the neighbours do), then lean and auto for the A/B.
# Everything lands in gpurun_out/first_contact/.
set -u
STEPS=${1:-20}
OUT=gpurun_out/first_contact
mkdir -p "$OUT"
NGPU=$(python - <<'PY'
import ctypes, os
try:
    lib = ctypes.CDLL(os.path.join("topo_descriptors_amd", "libtopo_amd.so"))
    print(lib.topo_amd_device_count())
except OSError:
    print(0)
PY
)
echo "GPUs visible: $NGPU" | tee "$OUT/summary.txt"
for N in 2 4 8; do
  if [ "$N" -gt "$NGPU" ]; then echo "skip N=$N (only $NGPU GPUs)" | tee -a "$OUT/summary.txt"; continue; fi
  for MODE in careful lean auto; do
    PORT=$((29500 + N * 10 + ${#MODE}))
    LOG="$OUT/n${N}_${MODE}.log"
    echo "== N=$N gate=$MODE" | tee -a "$OUT/summary.txt"
    HSA_ENABLE_IPC_MODE_LEGACY=0 NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,P2P TOPO_AMD_GATE_MODE=$MODE \
      timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$PORT" \
      bench.py --gpus "$N" --steps "$STEPS" --warmup 5 > "$LOG" 2>&1
    echo "   exit code $?" | tee -a "$OUT/summary.txt"
    # the bench line (last line that parses as JSON) - value, ms per step, gate give-ups, shard table
    python - "$LOG" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
line = None
for l in open(sys.argv[1], errors="replace"):
    l = l.strip()
    if l.startswith("{") and l.endswith("}"):
        try:
            line = json.loads(l)
        except ValueError:
            pass
if line is None:
    print("   no bench line (see the log)")
else:
    print(f"   value {line.get('value')} {line.get('unit')}  ms/step {line.get('ms_per_step')}  n_gpus {line.get('n_gpus')}  "
          f"gate_giveups {line.get('gate_giveups')}")
    for k, v in (line.get("descriptors") or {}).items():
        if isinstance(v, dict) and "ms" in v:
            print(f"      {k:34s} {v['ms']:9.4f} ms" + (f"  shard efficiency {v['shard_efficiency']}" if "shard_efficiency" in v else ""))
PY
    # channels RCCL set up (NCCL INFO lines name them per peer)
    echo "   RCCL channel lines: $(grep -c -i 'channel' "$LOG")  (first three below)" | tee -a "$OUT/summary.txt"
    grep -i -m 3 'channel' "$LOG" | sed 's/^/      /' | tee -a "$OUT/summary.txt"
    grep -i -m 3 'nChannels\|channels per' "$LOG" | sed 's/^/      /' | tee -a "$OUT/summary.txt"
  done
done
echo "done: $OUT/summary.txt"
