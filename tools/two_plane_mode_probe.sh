#!/bin/bash
# The two modes of the two-plane kernels (TPI + STD at 7 px: 2.8 or 3.7 ms on 32768^2, one mode per PROCESS): six processes
# under rocprofv3 with the translation counters of the vector L1 - does the slow mode come with translation misses?
# usage (GPU box, repo root): tools/two_plane_mode_probe.sh > gpurun_out/two_plane_mode.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do
  rm -rf /tmp/tp$i
  rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE --output-format csv -d /tmp/tp$i -- python3 $R/tools/std_trace.py 32768 7 > /dev/null 2>&1
  python3 - $i <<'PY'
import csv, glob, sys, collections
i = sys.argv[1]
t = collections.defaultdict(list)
for f in glob.glob(f"/tmp/tp{i}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "std_ring_spec_kernel<7, true, false>" in r["Kernel_Name"]:
            t["ms"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
c = collections.defaultdict(list)
for f in glob.glob(f"/tmp/tp{i}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "std_ring_spec_kernel<7, true, false>" in r["Kernel_Name"]:
            c[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"process {i}: TPI+STD 7 px kernel ms {[round(x, 3) for x in t['ms']]}  " + "  ".join(f"{k} {sum(v)/len(v):.4g}" for k, v in sorted(c.items())))
PY
done
