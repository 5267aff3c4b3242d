R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/s1/tr
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trx
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trx -o g -- python3 $R/tools/grad_trace.py 32768 3.25 > /dev/null 2>&1
python3 - /tmp/trx <<'PY' > $R/gpurun_out/s1/tr/other.txt
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r["Kernel_Name"].replace("void topo::(anonymous namespace)::", "")[:70]
    agg[n][0] += 1; agg[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for n, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]): print(f"{n:72s} launches {c:4d}  total {us/6:9.1f} us per call  mean {us/c:8.1f} us")
PY
python3 $R/tools/trace_window.py /tmp/trx 70 gradient_epilogue4_if > $R/gpurun_out/s1/tr/window.txt 2>&1
