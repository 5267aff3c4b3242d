#!/bin/bash
# Write-traffic A/B of the TPI kernel on the same 32768^2 integer DEM (tools/ubench/tpi_write_ab.hip):
# variant 0 = the product's pair (scratch-free fast build + general build over deferred tiles),
# 12 = general 12-wave build alone (spills), 8 = general 8-wave build (no scratch).  One counter per
# pass.  Also times variants 0 and 12 on a DEM with fractional elevations.
# usage: tools/ab_tpi_write.sh <outdir-under-gpurun_out>
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 0 12 8; do
  $R/tools/ubench/tpi_write_ab $v 32768 5 > $OUT/plain_v$v.json 2> $OUT/plain_v$v.err
  for c in WRITE_SIZE FETCH_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/v${v}_$c -- $R/tools/ubench/tpi_write_ab $v 32768 3 > $OUT/v${v}_$c.log 2>&1
  done
done
for v in 0 12; do
  $R/tools/ubench/tpi_write_ab $v 32768 5 0 > $OUT/plain_frac_v$v.json 2> $OUT/plain_frac_v$v.err
done
python3 - "$OUT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
res = {}
for v in (0, 12, 8):
    r = {"timing": json.loads(open(f"{out}/plain_v{v}.json").read().strip() or "{}")}
    for c in ("WRITE_SIZE", "FETCH_SIZE"):
        vals = []
        for f in glob.glob(f"{out}/v{v}_{c}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if "disc_wave_kernel" in row["Kernel_Name"] and row["Counter_Name"] == c:
                    vals.append((row["Kernel_Name"].split("disc_wave_kernel")[1][:32], float(row["Counter_Value"])))
        r[c + "_KiB_per_launch"] = vals
    if v != 8:
        r["timing_fractional_dem"] = json.loads(open(f"{out}/plain_frac_v{v}.json").read().strip() or "{}")
    res["product_pair" if v == 0 else f"waves{v}"] = r
json.dump(res, open(out + "/ab.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
