#!/bin/bash
# The profiles a round commits: rocprofv3 --kernel-trace --stats of the bench.py timed region (overlapped pipelines,
# and with every stage on one stream), then the three counter passes (SQ; FETCH_SIZE; WRITE_SIZE -- each in a run
# of its own).  Run on the GPU box: tools/profile_round.sh r02x ; results under gpurun_out/<tag>_*.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-prof}
ARGS="--steps 20 --warmup 2 --no-secondary --no-cpu-baseline"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_kt -o kt -- python3 $R/bench.py $ARGS > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err || exit 1
ZG16_SERIAL=1 timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_kt_serial -o kt -- python3 $R/bench.py $ARGS > $R/gpurun_out/${TAG}_bench_serial.json 2> $R/gpurun_out/${TAG}_bench_serial.err || exit 1
PARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/${TAG}_pmc_sq -o sq -- python3 $R/bench.py $PARGS > $R/gpurun_out/${TAG}_pmc_sq.log 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_f -o f -- python3 $R/bench.py $PARGS > $R/gpurun_out/${TAG}_pmc_f.log 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_w -o w -- python3 $R/bench.py $PARGS > $R/gpurun_out/${TAG}_pmc_w.log 2>&1 || exit 1
one() { ls $1/*/*$2 $1/*$2 2>/dev/null | head -1; }
python3 $R/tools/pmc_summary.py $(one $R/gpurun_out/${TAG}_pmc_f counter_collection.csv) $(one $R/gpurun_out/${TAG}_pmc_w counter_collection.csv) $R/gpurun_out/${TAG}_pmc_hbm.json > $R/gpurun_out/${TAG}_pmc_hbm.txt
python3 $R/tools/pmc_sq_summary.py $(one $R/gpurun_out/${TAG}_pmc_sq counter_collection.csv) $R/gpurun_out/${TAG}_pmc_sq.json > $R/gpurun_out/${TAG}_pmc_sq.txt
cp $(one $R/gpurun_out/${TAG}_kt kernel_stats.csv) $R/gpurun_out/${TAG}_kernel_stats.csv
cp $(one $R/gpurun_out/${TAG}_kt_serial kernel_stats.csv) $R/gpurun_out/${TAG}_serial_kernel_stats.csv
echo profiles done
