R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/s1/tr
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do
  rm -rf /tmp/tr$i
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr$i -o g -- python3 $R/tools/grad_trace.py 32768 3.25 > /dev/null 2>&1
  python3 - /tmp/tr$i <<'PY' >> $R/gpurun_out/s1/tr/summary.txt
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(float)
for r in rows:
    n = r["Kernel_Name"]
    key = "fused" if "fused" in n else "epilogue" if "epilogue" in n else "other"
    agg[key] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
t0 = min(int(r["Start_Timestamp"]) for r in rows if "fused" in r["Kernel_Name"]); t1 = max(int(r["End_Timestamp"]) for r in rows)
print({k: round(v / 6, 3) for k, v in agg.items()}, "span_ms_per_call", round((t1 - t0) / 1e6 / 6, 3))
PY
done
