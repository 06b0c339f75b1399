#!/bin/bash
# Per-kernel average durations of the bench.py timed region with every stage on one stream (ZG16_SERIAL=1), as a
# short table: tools/kstats.sh <tag> [bench args]; result in gpurun_out/<tag>_serial_kernel_stats.csv and on stdout.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-ks}
shift
ARGS=${*:-"--steps 20 --warmup 2 --no-secondary --no-cpu-baseline"}
cd /tmp && export TMPDIR=/tmp
ZG16_SERIAL=1 timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_kt_serial -o kt -- python3 $R/bench.py $ARGS > $R/gpurun_out/${TAG}_bench_serial.json 2> $R/gpurun_out/${TAG}_bench_serial.err || exit 1
f=$(ls $R/gpurun_out/${TAG}_kt_serial/*/*kernel_stats.csv $R/gpurun_out/${TAG}_kt_serial/*kernel_stats.csv 2>/dev/null | head -1)
cp $f $R/gpurun_out/${TAG}_serial_kernel_stats.csv
python3 - $f <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Name"]).replace("void zg::", "").replace("zg::", "")
    name = name.replace("Fp<FqParams> ", "G1").replace("Fp<zg::FqParams> ", "G1")
    print("%-40s calls %5s avg %9.1f us  total %8.2f ms" % (name[:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
