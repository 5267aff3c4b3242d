cd /root/repo
for v in "$@"; do
  echo "== $v"
  TOPO_AMD_LIBRARY=/root/repo/lab_libs/libtopo_$v.so VM_TIME_ONLY=0 timeout 300 python tools/valley_mfma_check.py 8192 2>&1 | tail -1
done
echo "== HEAD library"; VM_TIME_ONLY=0 timeout 300 python tools/valley_mfma_check.py 8192 2>&1 | tail -1
