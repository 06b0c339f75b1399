"""Per-kernel VALU issue utilisation from one rocprofv3 --pmc pass over the SQ counters.

    python tools/pmc_sq_summary.py <counter_collection.csv> <out.json>

Pass: rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU
-- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline.  Counter collection serialises the streams, so the
durations are uncontended.  valu_issue_utilisation = wave-level VALU instructions / (1024 SIMDs x duration in
4-clock issue slots at 2.4 GHz); v_mad_u64_u32 occupies about 1.5 slots, so a multiply-heavy kernel saturates
near 0.75-0.8.
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


def main(path, out):
    acc = defaultdict(lambda: defaultdict(float))
    seen = defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen[k]:
            seen[k].add(r["Dispatch_Id"])
            acc[k]["_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    res = {}
    for k, c in sorted(acc.items()):
        n = len(seen[k])
        if not k.startswith("k_"):
            continue
        us = c["_ns"] / n / 1e3
        valu = c["SQ_INSTS_VALU"] / n
        slots = 1024 * us * 1e-6 * 2.4e9 / 4
        res[k] = {"launches_sampled": n, "avg_duration_us": round(us, 1), "valu_insts_per_launch": valu,
                  "salu_insts_per_launch": c["SQ_INSTS_SALU"] / n, "waves_per_launch": c["SQ_WAVES"] / n,
                  "wave_quad_cycles_per_launch": c["SQ_WAVE_CYCLES"] / n,
                  "valu_issue_utilisation": round(valu / slots, 3) if slots else None}
    json.dump({"note": __doc__.split("Pass: ")[1].strip().replace("\n", " "), "kernels": res}, open(out, "w"), indent=1)
    for k, v in res.items():
        print("%-36s %8.1f us  valu util %.3f  (%d launches)" % (k[:36], v["avg_duration_us"], v["valu_issue_utilisation"] or 0, v["launches_sampled"]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
