#!/bin/bash
# Collect rocprofv3 PMC counters for one command, one counter group per pass (TCC FETCH/WRITE
# cannot share a pass; never combined with tracing other than --kernel-trace).
# usage: tools/pmc_passes.sh <outdir-under-gpurun_out> <args of the script...>
#        PMC_SCRIPT=tools/std_trace.py tools/pmc_passes.sh <outdir> <args>   (default script: bench.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" ${PMC_EXTRA:+"$PMC_EXTRA"}; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pass$i -- python3 $R/${PMC_SCRIPT:-bench.py} "$@" > $OUT/pass$i.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:90]][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k, cs in agg.items():
        fh.write(k + "\n")
        for c, v in sorted(cs.items()):
            fh.write(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}\n")
print(open(out + "/summary.txt").read()[:6000])
PY
