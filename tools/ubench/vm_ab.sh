# A/B of builds of the library on the valley index (7 px, 8192^2): tools/ubench/vm_ab.sh <tag> ... with lab_libs/libtopo_<tag>.so
cd /root/repo
for v in "$@"; do
  echo "== $v"
  TOPO_AMD_LIBRARY=/root/repo/lab_libs/libtopo_$v.so VM_TIME_ONLY=0 timeout 300 python tools/valley_mfma_check.py 8192 2>&1 | tail -1
done
echo "== the library in the tree"; VM_TIME_ONLY=0 timeout 300 python tools/valley_mfma_check.py 8192 2>&1 | tail -1
