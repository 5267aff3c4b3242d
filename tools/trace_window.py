"""Print a window of a rocprofv3 kernel trace (kernel_trace.csv under <dir>): the launches around the middle launch whose
name contains <pattern>, with start / end in microseconds.  python tools/trace_window.py <dir> <count> [pattern]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
pattern = sys.argv[3] if len(sys.argv) > 3 else "tpi_march"
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if pattern in r["Kernel_Name"]]
i0 = max(0, idx[len(idx) // 2] - 2)
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + int(sys.argv[2])]:
    name = r["Kernel_Name"].replace("void topo::(anonymous namespace)::", "")
    print(name[:56].ljust(56), r["Stream_Id"], round((int(r["Start_Timestamp"]) - t0) / 1e3, 1),
          round((int(r["End_Timestamp"]) - t0) / 1e3, 1), r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"], r["LDS_Block_Size"])
