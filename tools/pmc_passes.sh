#!/bin/bash
# Three separate rocprofv3 --pmc passes over a short bench.py run (SQ counters; FETCH_SIZE; WRITE_SIZE).
# Run from anywhere on the GPU box; writes under gpurun_out/pmc_*.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc_sq -o sq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_sq.log 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_f -o f -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_f.log 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_w -o w -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_w.log 2>&1 || exit 1
echo passes done
