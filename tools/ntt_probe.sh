#!/bin/bash
# Stand-alone transform timings (2^20 .. 2^26) and the HBM counter passes of a 2^20 transform.
# Run on the GPU box from the repo root; writes under gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-probe}
for n in 20 22 24 26; do
  timeout -k 10 120 python3 $R/tools/bench_standalone.py --log-n $n --skip-msm --reps 5 >> $R/gpurun_out/ntt_$TAG.jsonl || exit 1
done
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ntt_${TAG}_kt -o kt -- python3 $R/tools/bench_standalone.py --log-n 20 --skip-msm --reps 5 > /dev/null 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/ntt_${TAG}_f -o f -- python3 $R/tools/bench_standalone.py --log-n 20 --skip-msm --reps 5 > /dev/null 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/ntt_${TAG}_w -o w -- python3 $R/tools/bench_standalone.py --log-n 20 --skip-msm --reps 5 > /dev/null 2>&1 || exit 1
python3 $R/tools/pmc_summary.py $(ls $R/gpurun_out/ntt_${TAG}_f/*/*counter_collection.csv $R/gpurun_out/ntt_${TAG}_f/*counter_collection.csv 2>/dev/null | head -1) $(ls $R/gpurun_out/ntt_${TAG}_w/*/*counter_collection.csv $R/gpurun_out/ntt_${TAG}_w/*counter_collection.csv 2>/dev/null | head -1) $R/gpurun_out/ntt_${TAG}_pmc.json | grep -i ntt
grep -h k_ntt_pass $(ls $R/gpurun_out/ntt_${TAG}_kt/*/*kernel_stats.csv $R/gpurun_out/ntt_${TAG}_kt/*kernel_stats.csv 2>/dev/null | head -1)
cat $R/gpurun_out/ntt_$TAG.jsonl
