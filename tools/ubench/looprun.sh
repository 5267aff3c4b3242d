for r in 16 8 4 0 32; do
TOPO_AMD_RESERVE_CUS=$r TOPO_AMD_HALO_LOOPBACK=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --ny 4096 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('reserve $r', r['ms_per_step'], {k:v['ms'] for k,v in r['descriptors'].items()})"
done
