for r in 1 2; do
python tools/grad_time.py 3.25 30.25 | sed "s/^/plain stores: /"
TOPO_AMD_LIBRARY=$PWD/topo_descriptors_amd/libtopo_amd_nt.so python tools/grad_time.py 3.25 30.25 | sed "s/^/nontemporal: /"
done
