#!/bin/bash
# Write-traffic A/B of the one-pass TPI kernel: 12-wave (spilling) vs 8-wave (no scratch) build of the
# same kernel on the same 32768^2 integer DEM (tools/ubench/tpi_write_ab.hip).  One counter per pass.
# usage: tools/ab_tpi_write.sh <outdir-under-gpurun_out>
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 12 8; do
  $R/tools/ubench/tpi_write_ab $v 32768 5 > $OUT/plain_v$v.json 2> $OUT/plain_v$v.err
  for c in WRITE_SIZE FETCH_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/v${v}_$c -- $R/tools/ubench/tpi_write_ab $v 32768 3 > $OUT/v${v}_$c.log 2>&1
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
res = {}
for v in (12, 8):
    r = {"timing": json.loads(open(f"{out}/plain_v{v}.json").read().strip() or "{}")}
    for c in ("WRITE_SIZE", "FETCH_SIZE"):
        vals = []
        for f in glob.glob(f"{out}/v{v}_{c}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if "disc_wave_kernel" in row["Kernel_Name"] and row["Counter_Name"] == c:
                    vals.append(float(row["Counter_Value"]))
        r[c + "_KiB_per_launch"] = vals
    res[f"waves{v}"] = r
json.dump(res, open(out + "/ab.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
