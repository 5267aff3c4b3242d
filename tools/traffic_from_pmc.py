#!/usr/bin/env python3
"""HBM traffic per launch of one kernel from rocprofv3 PMC passes (tools/pmc_passes.sh output).

Follows /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads (16 B/lane, which is how
the disc kernel loads), so it is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.
The result records the git head (argument 4, the GPU box has no .git) and the hash of the kernel sources the
profile was taken on; bench.py nulls roofline.traffic when the sources have changed since.
Usage: tools/traffic_from_pmc.py <pmc dir> <kernel substring> <out.json> [git head]"""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


MATCHED = set()


def mean_counter(pmc_dir, kernel, counter):
    """Mean of `counter` over the launches whose kernel name contains `kernel` - the name as far as given, template
    arguments included ("std_ring_kernel<67, false" and "std_ring_kernel<67, true" are different kernels: round 3's
    r03_std67_traffic.json averaged the two).  The distinct names matched are recorded; more than one is an error."""
    vals = []
    for f in glob.glob(pmc_dir + "/pass*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if kernel in row["Kernel_Name"] and row["Counter_Name"] == counter:
                vals.append(float(row["Counter_Value"]))
                MATCHED.add(row["Kernel_Name"].replace("(anonymous namespace)", "").split("(")[0])
    return sum(vals) / len(vals) if vals else None


def main():
    pmc_dir, kernel, out = sys.argv[1:4]
    import bench  # noqa: PLC0415  (the hash bench.py compares with)
    fetch = mean_counter(pmc_dir, kernel, "FETCH_SIZE")
    write = mean_counter(pmc_dir, kernel, "WRITE_SIZE")
    hit = mean_counter(pmc_dir, kernel, "TCC_HIT_sum")
    miss = mean_counter(pmc_dir, kernel, "TCC_MISS_sum")
    if len(MATCHED) > 1:
        sys.exit("traffic_from_pmc: '%s' matches %d kernels, name one of them in full: %s" % (kernel, len(MATCHED), sorted(MATCHED)))
    result = {
        "kernel": kernel,
        "kernel_matched": sorted(MATCHED),
        "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
        "fetch_bytes_corrected": None if fetch is None else fetch * 1024 * 2,
        "write_bytes": None if write is None else write * 1024,
        "traffic_bytes_per_launch": None if fetch is None or write is None else fetch * 2048 + write * 1024,
        "l2_hit_rate": None if not hit else hit / (hit + miss),
        "correction": "FETCH_SIZE x2 (gfx950, 16 B/lane coalesced loads), WRITE_SIZE x1; KiB -> bytes",
        "git_head": sys.argv[4] if len(sys.argv) > 4 else None,
        "kernel_sources_sha256": bench.kernel_sources_sha256(),
    }
    json.dump(result, open(out, "w"), indent=1)
    print(json.dumps(result))


if __name__ == "__main__":
    main()
