"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs).

    python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>

Values are per launch.  rocprofv3 reports both counters in KiB.  On gfx950 FETCH_SIZE tallies a
128-byte request as 64 bytes for wide coalesced reads, so it is doubled; WRITE_SIZE is exact for
16-byte-per-lane stores.  Other access widths are uncalibrated: for the gather kernels
(k_accumulate reads one 64/128-byte base per entry) the raw figure already matches the byte count,
so both the raw and the doubled fetch figure are kept.
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name).replace("zg::", "")
    name = re.sub(r"Fp<FqParams\s*>", "Fq", name)
    return name.split("(")[0]


def load(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        acc[k][0] += float(r["Counter_Value"])
        acc[k][1] += 1
    return acc


def main(fetch_csv, write_csv, out):
    f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) | set(w)):
        raw = 1024 * f[k][0] / max(f[k][1], 1)
        wb = 1024 * w[k][0] / max(w[k][1], 1)
        res[k] = {"launches_sampled": f[k][1], "fetch_bytes_per_launch_raw": raw, "fetch_bytes_per_launch": 2 * raw,
                  "write_bytes_per_launch": wb, "hbm_bytes_per_launch": 2 * raw + wb}
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 3 --warmup 1`; "
                       "KiB -> bytes, FETCH_SIZE doubled (gfx950 counts a 128-byte read request as 64 bytes)",
               "kernels": res}, open(out, "w"), indent=1)
    for k, v in res.items():
        print("%-40s fetch %12.0f  write %12.0f  B/launch  (%d launches)" % (k[:40], v["fetch_bytes_per_launch"],
                                                                          v["write_bytes_per_launch"], v["launches_sampled"]))


if __name__ == "__main__":
    main(*sys.argv[1:4])
