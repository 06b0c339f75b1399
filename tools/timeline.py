"""Timeline of one prove step from a rocprofv3 --kernel-trace csv: kernels per queue with start/end in us.

    python tools/timeline.py <kernel_trace.csv> [step_from_end]
A step starts at the first k_spmv_* launch; steps are counted from the end of the trace.
"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name).replace("zg::", "")
    name = re.sub(r"Fp<FqParams\s*>", "Fq", name)
    return name.split("(")[0]


def main(path, back=1):
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"]) for r in csv.DictReader(open(path))]
    rows.sort()
    # step boundaries: a k_spmv_* launch more than 0.1 ms after the previous one opens a prove
    starts, last = [], -10**18
    for i, r in enumerate(rows):
        if r[2].startswith("k_spmv"):
            if r[0] - last > 100000:
                starts.append(i)
            last = r[0]
    i0 = starts[-back]
    i1 = starts[-back + 1] if back > 1 else len(rows)
    t0 = rows[i0][0]
    queues = sorted({r[3] for r in rows[i0:i1]})
    for s, e, k, q in rows[i0:i1]:
        print("%8.1f %8.1f  %7.1f  q%-2d %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, queues.index(q), k))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
