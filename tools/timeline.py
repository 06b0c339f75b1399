#!/usr/bin/env python3
"""Timeline of the last proof in a rocprofv3 --kernel-trace CSV: every kernel's start and duration relative to the
proof's first kernel, by hardware queue, plus how many microseconds of the proof had 0, 1, 2, ... kernels running.
usage: timeline.py kernel_trace.csv [first-kernel-substring]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else "k_spmv_short"
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows]
ks.sort()
starts = [i for i, k in enumerate(ks) if first in k[2]]
# a proof begins shortly before its row evaluation (the witness multiexps start first): take the window between the
# last two row evaluations
lo_i, hi_i = starts[-2], starts[-1]
t_lo = ks[lo_i][0]
window = [k for k in ks if t_lo - 300_000 <= k[0] < ks[hi_i][0] - 300_000]
t0 = min(k[0] for k in window)


def short(name):
    name = re.sub(r"\(.*", "", name).replace("void zg::", "").replace("zg::", "")
    return name.replace("Fp<FqParams> ", "G1").replace("Fp<FqParams>", "G1")[:30]


queues = sorted({k[3] for k in window})
print("queues:", queues)
for s, e, n, q in window:
    print("%9.1f us  +%8.1f us  q%-3s %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, q, "    " * queues.index(q), short(n)))
events = sorted([(s, 1) for s, e, n, q in window] + [(e, -1) for s, e, n, q in window])
depth, last, hist = 0, t0, {}
for t, d in events:
    hist[depth] = hist.get(depth, 0) + (t - last)
    depth, last = depth + d, t
print("span %.1f us; time with n kernels running:" % ((last - t0) / 1e3),
      {n: round(v / 1e3, 1) for n, v in sorted(hist.items())})
